"""Device-resident graph maps for a batch of episodes.

`BatchedGraphMap` replaces, for inference rollouts, the per-episode `GraphMap` objects of the reference
(finetune_src/models/graph_utils.py:95-148), the graph / node-embedding updates of the rollout loop
(finetune_src/r2r/agent.py:406-408, 434-451, 565-569) and the builders `_nav_gmap_variable` / `_nav_vp_variable`
(agent.py:157-278).  The host keeps only what needs strings: the viewpoint-id -> slot dictionary of every episode (slot =
insertion order into the reference's `node_positions`, so slot 0 is the start viewpoint) and which slots are visited (to
hand the agent its `gmap_vpids` lists).  Everything numeric lives on the GPU and is touched by three kernels per step
(graphmap.hip): observe -> update_embeds -> nav_inputs.  Embeddings are stored detached: this is the rollout path of
evaluation / serving, not of back-propagation through the graph map.

    gm = BatchedGraphMap(B, hidden, device, dtype)
    gm.reset(obs)                                   # rollout start: update_graph(ob) of every episode
    for t in range(max_action_len):
        gm.begin_step(obs, t, ended)                # node_step_ids[cur] = t + 1
        pano_embeds, pano_masks, pano_fused = model('panorama', pano_inputs)
        gm.update_embeds(pano_embeds, pano_fused, ended)
        nav_inputs = gm.nav_inputs(pano_embeds, nav_types, view_lens, last_embeds)
        ... model('navigation', nav_inputs) ... env step ...
        gm.observe(new_obs, ended)                  # update_graph(ob) of the episodes that are still running
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, kernels as K


class BatchedGraphMap:
    def __init__(self, batch_size, hidden_size, device, dtype=torch.bfloat16, max_nodes=64, max_cand=16):
        if max_nodes > 128:
            raise _lib.DsrgHipError('BatchedGraphMap: at most 128 nodes per episode')
        self.B, self.H, self.N, self.C = batch_size, hidden_size, max_nodes, max_cand
        self.device, self.dtype = torch.device(device), dtype
        B, N, H, dev = batch_size, max_nodes, hidden_size, self.device
        self.pos = torch.empty(B, N, 3, device=dev, dtype=torch.float64)
        self.dist = torch.empty(B, N, N, device=dev, dtype=torch.float64)
        self.point = torch.empty(B, N, N, device=dev, dtype=torch.int32)
        self.visited = torch.empty(B, N, device=dev, dtype=torch.uint8)
        self.step_id = torch.empty(B, N, device=dev, dtype=torch.int32)
        self.emb_sum = torch.empty(B, N, H, device=dev, dtype=torch.float32)
        self.emb_cnt = torch.empty(B, N, device=dev, dtype=torch.int32)
        self.n_nodes = torch.empty(B, device=dev, dtype=torch.int32)
        self._state = _lib.GraphState(*[t.data_ptr() for t in (self.pos, self.dist, self.point, self.visited, self.step_id,
                                                                self.emb_sum, self.emb_cnt, self.n_nodes)], N, H)
        self.slots = None          # per episode: {viewpoint id: slot}
        self.names = None          # per episode: [viewpoint id of slot 0, 1, ...]
        self.host_visited = None   # per episode: set of visited slots
        self._cur = None           # device tensors of the last observation (slots, heading, ...)

    # ------------------------------------------------------------------ host bookkeeping + packing of one observation
    def _pack(self, obs, active):
        B, Cm = self.B, self.C
        # two packed host buffers -> two H2D copies: int32 [cur | ncand | n_new | cand], float64 [cpos | head | elev | kpos]
        ib = np.zeros(B * (3 + Cm), np.int32)
        fb = np.zeros(B * (5 + 3 * Cm), np.float64)
        cur, ncand, n_new, cand = ib[:B], ib[B:2 * B], ib[2 * B:3 * B], ib[3 * B:].reshape(B, Cm)
        cpos, head, elev, kpos = fb[:3 * B].reshape(B, 3), fb[3 * B:4 * B], fb[4 * B:5 * B], fb[5 * B:].reshape(B, Cm, 3)
        for i, ob in enumerate(obs):
            sl, nm = self.slots[i], self.names[i]
            cc = ob['candidate']
            if len(cc) > Cm:
                raise _lib.DsrgHipError('BatchedGraphMap: %d candidates (max_cand=%d)' % (len(cc), Cm))
            if active[i]:
                for vp in [ob['viewpoint']] + [c['viewpointId'] for c in cc]:       # insertion order of node_positions
                    if vp not in sl:
                        sl[vp] = len(nm)
                        nm.append(vp)
                if len(nm) > self.N:
                    raise _lib.DsrgHipError('BatchedGraphMap: episode %d exceeds max_nodes=%d' % (i, self.N))
            cur[i] = sl[ob['viewpoint']]             # (an ended episode repeats its last observation: all ids are known)
            ncand[i] = len(cc)
            n_new[i] = len(nm)
            cpos[i] = ob['position']
            head[i], elev[i] = ob['heading'], ob['elevation']
            for j, c in enumerate(cc):
                cand[i, j] = sl[c['viewpointId']]
                kpos[i, j] = c['position']
        dev = self.device
        ibd, fbd = torch.from_numpy(ib).to(dev), torch.from_numpy(fb).to(dev)
        self._cur = {'cur': ibd[:B], 'ncand': ibd[B:2 * B], 'n_new': ibd[2 * B:3 * B], 'cand': ibd[3 * B:], 'cpos': fbd[:3 * B],
                     'head': fbd[3 * B:4 * B], 'elev': fbd[4 * B:5 * B], 'kpos': fbd[5 * B:], 'cur_np': cur, 'cand_np': cand,
                     'ncand_np': ncand}
        return self._cur

    def _observe(self, obs, graph_active, step_active, step_value):
        ga = np.asarray(graph_active, bool)
        p = self._pack(obs, ga)
        for i in np.nonzero(ga)[0]:
            self.host_visited[i].add(int(p['cur_np'][i]))
        m = torch.from_numpy(np.concatenate([ga, np.asarray(step_active, bool)]).astype(np.uint8)).to(self.device)
        gad, sad = m[:self.B], m[self.B:]
        _lib.check(_lib.load().dsrg_graph_observe(C.byref(self._state), K._p(gad), K._p(sad), K._p(p['cur']), K._p(p['n_new']), K._p(p['ncand']),
                                                  K._p(p['cand']), K._p(p['cpos']), K._p(p['kpos']), self.C, int(step_value), self.B,
                                                  K._stream()), 'dsrg_graph_observe')

    # ------------------------------------------------------------------ the reference's call sites
    def reset(self, obs):
        """agent.py:406-408: gmaps = [GraphMap(ob['viewpoint'])]; gmap.update_graph(ob)"""
        self.slots = [dict() for _ in range(self.B)]
        self.names = [[] for _ in range(self.B)]
        self.host_visited = [set() for _ in range(self.B)]
        _lib.check(_lib.load().dsrg_graph_reset(*[K._p(t) for t in (self.pos, self.dist, self.point, self.visited, self.step_id, self.emb_sum,
                                                                    self.emb_cnt, self.n_nodes)], self.B, self.N, self.H, K._stream()),
                   'dsrg_graph_reset')
        self._observe(obs, [True] * self.B, [False] * self.B, 0)

    def observe(self, obs, ended):
        """agent.py:565-569: gmap.update_graph(ob) for the episodes that have not ended"""
        self._observe(obs, ~np.asarray(ended, bool), [False] * self.B, 0)

    def begin_step(self, obs, t, ended):
        """agent.py:434-437: node_step_ids[current viewpoint] = t + 1 (not ended)"""
        if self._cur is None:
            raise _lib.DsrgHipError('BatchedGraphMap: reset() first')
        p = self._cur
        m = torch.from_numpy(np.concatenate([np.zeros(self.B, bool), ~np.asarray(ended, bool)]).astype(np.uint8)).to(self.device)
        none, sa = m[:self.B], m[self.B:]
        _lib.check(_lib.load().dsrg_graph_observe(C.byref(self._state), K._p(none), K._p(sa), K._p(p['cur']), K._p(p['n_new']), K._p(p['ncand']),
                                                  K._p(p['cand']), K._p(p['cpos']), K._p(p['kpos']), self.C, int(t) + 1, self.B, K._stream()),
                   'dsrg_graph_observe')

    def update_embeds(self, pano_embeds, pano_fused, ended):
        """agent.py:443-451"""
        p = self._cur
        pe, pf = pano_embeds.detach(), pano_fused.detach()
        if pe.dtype != pf.dtype or pe.stride(2) != 1 or pf.stride(1) != 1:
            raise _lib.DsrgHipError('BatchedGraphMap.update_embeds: embeddings must share a dtype and be row-contiguous')
        act = torch.from_numpy((~np.asarray(ended, bool)).astype(np.uint8)).to(self.device)
        _lib.check(_lib.load().dsrg_graph_update_embeds(C.byref(self._state), K._p(act), K._p(p['cur']), K._p(p['ncand']), K._p(p['cand']), self.C,
                                                        K._p(pe), pe.stride(0), pe.stride(1), K._p(pf), pf.stride(0), self.B, K._dt(pe),
                                                        K._stream()), 'dsrg_graph_update_embeds')

    def gmap_vpids(self):
        """[None, None] + visited + unvisited viewpoint ids per episode (agent.py:165-181), from the host-side mirror"""
        out = []
        for nm, vis in zip(self.names, self.host_visited):
            out.append([None, None] + [v for s, v in enumerate(nm) if s in vis] + [v for s, v in enumerate(nm) if s not in vis])
        return out

    def current_token_index(self, nav):
        """int64 [B]: position of each episode's current node in the graph-map token order of `nav` (= nav_inputs() output);
        what GraphedNavStep.set_current_node() takes"""
        return (nav['gmap_slots'] == self._cur['cur'][:, None]).int().argmax(1)

    def nav_inputs(self, pano_embeds, nav_types, view_lens, last_embeds=None):
        """_nav_gmap_variable + _nav_vp_variable (agent.py:157-278) -> the dict forward(mode='navigation') takes (plus the device-side
        fusion indices `gmap_src` / `bw_mask`, so the model's forward does no host work)."""
        p, B, H, dev, dt = self._cur, self.B, self.H, self.device, pano_embeds.dtype
        V = pano_embeds.shape[1]
        Vp = V + 2
        vpids = self.gmap_vpids()
        G = max(len(x) for x in vpids)
        o = {'gmap_img_embeds': torch.empty(B, G, H, device=dev, dtype=dt), 'gmap_step_ids': torch.empty(B, G, device=dev, dtype=torch.int64),
             'gmap_pos_fts': torch.empty(B, G, 7, device=dev), 'gmap_pair_dists': torch.empty(B, G, G, device=dev),
             'gmap_masks': torch.empty(B, G, device=dev, dtype=torch.bool), 'gmap_visited_masks': torch.empty(B, G, device=dev, dtype=torch.bool),
             'gmap_slots': torch.empty(B, G, device=dev, dtype=torch.int32), 'gmap_src': torch.empty(B, G, device=dev, dtype=torch.int32),
             'bw_mask': torch.empty(B, Vp, device=dev, dtype=torch.uint8), 'vp_pos_fts': torch.empty(B, Vp, 14, device=dev),
             'no_vp_left': torch.empty(B, device=dev, dtype=torch.bool), 'gmap_len': torch.empty(B, device=dev, dtype=torch.int32)}
        out = _lib.GraphNavOut(*[o[k].data_ptr() for k in ('gmap_img_embeds', 'gmap_step_ids', 'gmap_pos_fts', 'gmap_pair_dists', 'gmap_masks',
                                                           'gmap_visited_masks', 'gmap_slots', 'gmap_src', 'bw_mask', 'vp_pos_fts',
                                                           'no_vp_left', 'gmap_len')], G, Vp)
        mem = None
        if last_embeds is not None:
            mem = last_embeds.detach().to(dt)
            mem = mem if mem.stride(1) == 1 else mem.contiguous()
        _lib.check(_lib.load().dsrg_graph_nav_inputs(C.byref(self._state), C.byref(out), K._p(p['cur']), K._p(p['head']), K._p(p['elev']),
                                                     K._p(p['ncand']), K._p(p['cand']), self.C, K._p(mem), mem.stride(0) if mem is not None else 0,
                                                     B, K._dt(o['gmap_img_embeds']), K._stream()), 'dsrg_graph_nav_inputs')
        # the viewpoint side that does not depend on the graph (agent.py:234-237, 262-267): plain tensor ops on the model's outputs
        pe = pano_embeds
        z = torch.zeros(B, 1, H, device=dev, dtype=dt)
        o['vp_img_embeds'] = torch.cat([z, z if mem is None else mem[:, None], pe], 1)
        o['vp_nav_masks'] = torch.cat([torch.ones(B, 1, dtype=torch.bool, device=dev), torch.zeros(B, 1, dtype=torch.bool, device=dev),
                                       nav_types == 1], 1)
        o['vp_masks'] = torch.arange(Vp, device=dev)[None] < (view_lens + 2)[:, None]
        o['vp_obj_masks'] = None
        o['gmap_vpids'] = vpids
        o['vp_cand_vpids'] = [[None, None] + [self.names[i][s] for s in p['cand_np'][i, :p['ncand_np'][i]]] for i in range(B)]
        return o
