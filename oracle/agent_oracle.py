"""TEST INFRASTRUCTURE ONLY (imported by tests/ and tools/make_golden_graph.py; never by dsrg_amd/).

CPU restatement of the agent-side navigation-input assembly of the reference:
  * GraphMap / FloydGraph            finetune_src/models/graph_utils.py:42-148   (pinned: tools/make_golden_graph.py drives
                                     the reference's own graph_utils.GraphMap, imported in the build container, through the
                                     same episodes and the fixture holds ITS results)
  * per-step graph / embedding update finetune_src/r2r/agent.py:406-408, 434-451, 565-569
  * _nav_gmap_variable               finetune_src/r2r/agent.py:157-229
  * _nav_vp_variable                 finetune_src/r2r/agent.py:231-278
The two builders live in agent.py, which cannot be imported here (MatterSim, h5py ... are absent): their restatement is
"parity unpinned" beyond what the pinned GraphMap feeds them; they are transcriptions of straight-line padding code.
Python dict / float64 semantics are kept exactly (insertion order = node order, double-precision distances)."""
from collections import defaultdict

import numpy as np

MAX_DIST = 30
MAX_STEP = 10
INF_DIST = 95959595


def calc_position_distance(a, b):            # graph_utils.py:7-13
    dx, dy, dz = b[0] - a[0], b[1] - a[1], b[2] - a[2]
    return np.sqrt(dx ** 2 + dy ** 2 + dz ** 2)


def calculate_vp_rel_pos_fts(a, b, base_heading=0, base_elevation=0):     # graph_utils.py:15-32
    dx, dy, dz = b[0] - a[0], b[1] - a[1], b[2] - a[2]
    xy_dist = max(np.sqrt(dx ** 2 + dy ** 2), 1e-8)
    xyz_dist = max(np.sqrt(dx ** 2 + dy ** 2 + dz ** 2), 1e-8)
    heading = np.arcsin(dx / xy_dist)
    if b[1] < a[1]:
        heading = np.pi - heading
    heading -= base_heading
    elevation = np.arcsin(dz / xyz_dist) - base_elevation
    return heading, elevation, xyz_dist


def get_angle_fts(headings, elevations, angle_feat_size):                  # graph_utils.py:34-40
    ang = np.vstack([np.sin(headings), np.cos(headings), np.sin(elevations), np.cos(elevations)]).transpose().astype(np.float32)
    rep = angle_feat_size // 4
    return np.concatenate([ang] * rep, 1) if rep > 1 else ang


class FloydGraph:                                                          # graph_utils.py:43-93
    def __init__(self):
        self._dis = defaultdict(lambda: defaultdict(lambda: INF_DIST))
        self._point = defaultdict(lambda: defaultdict(lambda: ""))
        self._visited = set()

    def distance(self, x, y):
        return 0 if x == y else self._dis[x][y]

    def add_edge(self, x, y, dis):
        if dis < self._dis[x][y]:
            self._dis[x][y] = self._dis[y][x] = dis
            self._point[x][y] = self._point[y][x] = ""

    def update(self, k):
        for x in self._dis:
            for y in self._dis:
                if x != y and self._dis[x][k] + self._dis[k][y] < self._dis[x][y]:
                    self._dis[x][y] = self._dis[y][x] = self._dis[x][k] + self._dis[k][y]
                    self._point[x][y] = self._point[y][x] = k
        self._visited.add(k)

    def visited(self, k):
        return k in self._visited

    def path(self, x, y):
        if x == y:
            return []
        if self._point[x][y] == "":
            return [y]
        k = self._point[x][y]
        return self.path(x, k) + self.path(k, y)


class GraphMap:                                                            # graph_utils.py:96-148
    def __init__(self, start_vp):
        self.start_vp = start_vp
        self.node_positions = {}
        self.graph = FloydGraph()
        self.node_embeds = {}
        self.node_step_ids = {}

    def update_graph(self, ob):
        self.node_positions[ob['viewpoint']] = ob['position']
        for cc in ob['candidate']:
            self.node_positions[cc['viewpointId']] = cc['position']
            self.graph.add_edge(ob['viewpoint'], cc['viewpointId'], calc_position_distance(ob['position'], cc['position']))
        self.graph.update(ob['viewpoint'])

    def update_node_embed(self, vp, embed, rewrite=False):
        if rewrite or vp not in self.node_embeds:
            self.node_embeds[vp] = [embed, 1]
        else:
            self.node_embeds[vp][0] = self.node_embeds[vp][0] + embed
            self.node_embeds[vp][1] += 1

    def get_node_embed(self, vp):
        return self.node_embeds[vp][0] / self.node_embeds[vp][1]

    def get_pos_fts(self, cur_vp, gmap_vpids, cur_heading, cur_elevation, angle_feat_size=4):
        rel_angles, rel_dists = [], []
        for vp in gmap_vpids:
            if vp is None:
                rel_angles.append([0, 0])
                rel_dists.append([0, 0, 0])
            else:
                h, e, d = calculate_vp_rel_pos_fts(self.node_positions[cur_vp], self.node_positions[vp],
                                                   base_heading=cur_heading, base_elevation=cur_elevation)
                rel_angles.append([h, e])
                rel_dists.append([d / MAX_DIST, self.graph.distance(cur_vp, vp) / MAX_DIST,
                                  len(self.graph.path(cur_vp, vp)) / MAX_STEP])
        rel_angles = np.array(rel_angles).astype(np.float32)
        rel_dists = np.array(rel_dists).astype(np.float32)
        return np.concatenate([get_angle_fts(rel_angles[:, 0], rel_angles[:, 1], angle_feat_size), rel_dists], 1)


def _pad(rows, width=None, dtype=np.float32):
    """pad_tensors / pad_sequence: stack [Ti, ...] arrays into [B, Tmax, ...] with zeros"""
    T = max(r.shape[0] for r in rows) if width is None else width
    out = np.zeros((len(rows), T) + rows[0].shape[1:], dtype)
    for i, r in enumerate(rows):
        out[i, :r.shape[0]] = r
    return out


def step_begin(gmaps, obs, t, ended):
    """agent.py:434-437"""
    for i, g in enumerate(gmaps):
        if not ended[i]:
            g.node_step_ids[obs[i]['viewpoint']] = t + 1


def update_embeds(gmaps, obs, cand_vpids, pano_embeds, pano_fused, ended):
    """agent.py:443-451 (numpy embeddings)"""
    for i, g in enumerate(gmaps):
        if not ended[i]:
            g.update_node_embed(obs[i]['viewpoint'], pano_fused[i], rewrite=True)
            for j, vp in enumerate(cand_vpids[i]):
                if not g.graph.visited(vp):
                    g.update_node_embed(vp, pano_embeds[i, j])


def nav_gmap_variable(obs, gmaps, last_embeds=None):
    """agent.py:157-229 (act_visited_nodes=False), numpy outputs"""
    vpids_b, lens, embeds, steps, pos, pairs, vis, no_left = [], [], [], [], [], [], [], []
    for i, g in enumerate(gmaps):
        visited = [k for k in g.node_positions if g.graph.visited(k)]
        unvisited = [k for k in g.node_positions if not g.graph.visited(k)]
        no_left.append(len(unvisited) == 0)
        vp = [None, None] + visited + unvisited
        vmask = [0, 0] + [1] * len(visited) + [0] * len(unvisited)
        node_e = [g.get_node_embed(v) for v in vp[2:]]
        mem = np.zeros_like(node_e[0]) if last_embeds is None else last_embeds[i]
        embeds.append(np.stack([np.zeros_like(node_e[0]), mem] + node_e, 0))
        steps.append(np.asarray([g.node_step_ids.get(v, 0) for v in vp], np.int64))
        pos.append(g.get_pos_fts(obs[i]['viewpoint'], vp, obs[i]['heading'], obs[i]['elevation']))
        pd = np.zeros((len(vp), len(vp)), np.float32)
        for a in range(2, len(vp)):
            for b in range(a + 1, len(vp)):
                pd[a, b] = pd[b, a] = g.graph.distance(vp[a], vp[b])
        pairs.append(pd)
        vis.append(np.asarray(vmask, bool))
        vpids_b.append(vp)
        lens.append(len(vp))
    G = max(lens)
    masks = np.arange(G)[None] < np.asarray(lens)[:, None]
    masks[:, 1] = False
    pair = np.zeros((len(gmaps), G, G), np.float32)
    for i, pd in enumerate(pairs):
        pair[i, :lens[i], :lens[i]] = pd
    return {'gmap_vpids': vpids_b, 'gmap_img_embeds': _pad(embeds), 'gmap_step_ids': _pad(steps, dtype=np.int64),
            'gmap_pos_fts': _pad(pos), 'gmap_visited_masks': _pad(vis, dtype=bool), 'gmap_pair_dists': pair,
            'gmap_masks': masks, 'no_vp_left': no_left}


def nav_vp_variable(obs, gmaps, pano_embeds, cand_vpids, view_lens, nav_types, last_embeds=None):
    """agent.py:231-278, numpy outputs (object inputs of REVERIE left out)"""
    B, V, H = pano_embeds.shape
    mem = np.zeros((B, 1, H), pano_embeds.dtype) if last_embeds is None else last_embeds[:, None]
    vp_img = np.concatenate([np.zeros((B, 1, H), pano_embeds.dtype), mem, pano_embeds], 1)
    pos = []
    for i, g in enumerate(gmaps):
        cand = g.get_pos_fts(obs[i]['viewpoint'], cand_vpids[i], obs[i]['heading'], obs[i]['elevation'])
        start = g.get_pos_fts(obs[i]['viewpoint'], [g.start_vp], obs[i]['heading'], obs[i]['elevation'])
        p = np.zeros((V + 2, 14), np.float32)
        p[:, :7] = start
        p[2:len(cand) + 2, 7:] = cand
        pos.append(p)
    nav = np.concatenate([np.ones((B, 1), bool), np.zeros((B, 1), bool), nav_types == 1], 1)
    vp_masks = np.arange(V + 2)[None] < (np.asarray(view_lens) + 2)[:, None]
    return {'vp_img_embeds': vp_img, 'vp_pos_fts': np.stack(pos, 0), 'vp_masks': vp_masks, 'vp_nav_masks': nav,
            'vp_cand_vpids': [[None, None] + list(x) for x in cand_vpids]}
