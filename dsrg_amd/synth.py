"""Deterministic synthetic weights and inputs for the DSRG nav-policy hot path.

Everything here is a pure integer hash (splitmix64 over (key, flat index)), so the
same tensors are regenerated bit-for-bit under any numpy/torch version, on the
CPU container (golden generation) and on the GPU box (parity tests, bench).

Input layouts follow what the reference agent feeds the network:
  language   : finetune_src/r2r/agent.py:38-73
  panorama   : finetune_src/r2r/agent.py:75-137  (REVERIE: reverie/agent_obj.py:60-135)
  navigation : finetune_src/r2r/agent.py:157-278 (REVERIE: reverie/agent_obj.py:150-250)
  pretrain   : pretrain_src/data/tasks.py:196-243,446-497 (collators)
"""
import zlib
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _raw(key, n, seed):
    base = np.uint64((zlib.crc32(key.encode()) << 20) ^ (int(seed) << 52))
    with np.errstate(over='ignore'):
        idx = np.arange(n, dtype=np.uint64) + base
        return _splitmix64(idx)


def uniform(key, shape, seed=0, lo=-1.0, hi=1.0):
    """float32 uniform in [lo, hi) with 24-bit resolution (exact in fp32)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = (_raw(key, n, seed) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def integers(key, shape, lo, hi, seed=0):
    """int64 uniform in [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    r = (_raw(key, n, seed) >> np.uint64(11)).astype(np.uint64) % np.uint64(hi - lo)
    return (r.astype(np.int64) + lo).reshape(shape)


# --------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------
def synth_tensor_for(name, shape, seed=0):
    """Rule: 1-D '*.weight' = LayerNorm gamma (1 + 0.1u); other 1-D = bias (0.02u);
    2-D with fan_in <= 16 (pos/loc/bbox/sprel linears) = 0.3u; other 2-D = 0.02*sqrt(3)*u."""
    shape = tuple(shape)
    u = uniform('w:' + name, shape, seed)
    if len(shape) == 1:
        if name.endswith('weight'):
            return (1.0 + 0.1 * u).astype(np.float32)
        return (0.02 * u).astype(np.float32)
    fan_in = shape[-1]
    scale = 0.3 if fan_in <= 16 else 0.02 * np.sqrt(3.0)
    return (scale * u).astype(np.float32)


def synth_state_dict(shapes, seed=0):
    """shapes: mapping name -> shape (e.g. from model.state_dict()). Returns numpy dict."""
    return {k: synth_tensor_for(k, tuple(v), seed) for k, v in shapes.items()}


# --------------------------------------------------------------------------
# inputs (numpy; callers move them to torch / device)
# --------------------------------------------------------------------------
def _lens(key, B, lo, hi, seed, force_max=True):
    ln = integers(key, (B,), lo, hi + 1, seed)
    if force_max:
        ln[0] = hi
    return ln


def _angle_feats(key, shape, seed):
    """[sin h, cos h, sin e, cos e] rows."""
    h = uniform(key + ':h', shape, seed, -np.pi, np.pi)
    e = uniform(key + ':e', shape, seed, -0.6, 0.6)
    return np.stack([np.sin(h), np.cos(h), np.sin(e), np.cos(e)], -1).astype(np.float32)


def language_batch(B, L, seed=0, La=6, Lo=6, min_len=20):
    txt_lens = _lens('txt_lens', B, min(min_len, L), L, seed)
    ids = integers('txt_ids', (B, L), 1000, 30000, seed)
    ar = np.arange(L)[None]
    masks = ar < txt_lens[:, None]
    ids = np.where(masks, ids, 0)
    ids[:, 0] = 101
    act_lens = _lens('act_lens', B, 1, La, seed)
    obj_lens = _lens('obj_lens_txt', B, 1, Lo, seed)
    act_masks = np.arange(La)[None] < act_lens[:, None]
    obj_masks = np.arange(Lo)[None] < obj_lens[:, None]
    act_ids = np.where(act_masks, integers('act_ids', (B, La), 1, 37, seed), 0)
    obj_ids = np.where(obj_masks, integers('obj_ids', (B, Lo), 1, 44, seed), 0)
    return {
        'txt_ids': ids.astype(np.int64), 'txt_masks': masks, 'txt_lens': txt_lens,
        'act_txt_ids': act_ids.astype(np.int64), 'act_txt_masks': act_masks,
        'obj_txt_ids': obj_ids.astype(np.int64), 'obj_txt_masks': obj_masks,
    }


def _objv2(key, B, V, O, seed):
    obj_lens = integers(key + ':lens', (B, V), 0, O + 1, seed)
    obj_lens[0, 0] = O                      # gen_seq_masks needs max == O (vilmodel.py:711)
    valid = np.arange(O)[None, None] < obj_lens[..., None]
    fts = uniform(key + ':bbox', (B, V, O, 8), seed, 0.0, 1.0)
    fts[..., 3:7] = _angle_feats(key + ':ang', (B, V, O), seed)
    fts[..., 7] = integers(key + ':cat', (B, V, O), 1, 45, seed).astype(np.float32)
    fts = np.where(valid[..., None], fts, 0.0).astype(np.float32)
    return fts, obj_lens


def panorama_batch(B, V=36, O=3, seed=0, reverie=False, No=20):
    """R2R keys as forward() reads them (vilmodel.py:844-849); REVERIE keys vilmodel_reverie.py:859-865."""
    ncand = integers('ncand', (B,), 2, 7, seed)
    nav_types = (np.arange(V)[None] < ncand[:, None]).astype(np.int64)
    view = uniform('view_img_fts', (B, V, 768), seed)
    loc = np.concatenate([_angle_feats('loc', (B, V), seed), np.ones((B, V, 3), np.float32)], -1)
    objv2, objv2_lens = _objv2('objv2', B, V, O, seed)
    out = {'view_img_fts': view, 'view_lens': np.full((B,), V, np.int64), 'ncand': ncand}
    if not reverie:
        out.update({'obj_img_fts': objv2, 'obj_lens': objv2_lens, 'loc_fts': loc, 'nav_types': nav_types})
        return out
    obj_lens = _lens('objv1_lens', B, 0, No, seed)
    ovalid = np.arange(No)[None] < obj_lens[:, None]
    obj_fts = np.where(ovalid[..., None], uniform('objv1_fts', (B, No, 768), seed), 0.0).astype(np.float32)
    oloc = np.concatenate([_angle_feats('oloc', (B, No), seed), uniform('obox', (B, No, 3), seed, 0.0, 1.0)], -1)
    oloc = np.where(ovalid[..., None], oloc, 0.0).astype(np.float32)
    out.update({
        'objv2_img_fts': objv2, 'objv2_lens': objv2_lens, 'objv2_nav_types': nav_types,
        'obj_img_fts': obj_fts, 'obj_lens': obj_lens,
        'loc_fts': np.concatenate([loc, oloc], 1),
        'nav_types': np.concatenate([nav_types, np.where(ovalid, 2, 0)], 1).astype(np.int64),
    })
    return out


def navigation_batch(B, G=20, Vp=38, seed=0, ncand=None, nav_types=None, vp_lens=None,
                     reverie=False, min_g=10, all_valid=False):
    """Graph-map + viewpoint inputs (agent.py:157-278). Token order [stop],[MEM],visited,unvisited.
    ncand / nav_types [B,Vp-2] / vp_lens [B] come from the matching panorama_batch when given."""
    if ncand is None:
        ncand = integers('ncand', (B,), 2, 7, seed)
    if nav_types is None:
        nav_types = (np.arange(Vp - 2)[None] < ncand[:, None]).astype(np.int64)
    if vp_lens is None:
        vp_lens = np.full((B,), Vp, np.int64)
    glen = np.full((B,), G, np.int64) if all_valid else _lens('gmap_lens', B, min(min_g, G), G, seed)
    nvis = integers('nvis', (B,), 1, 4, seed)
    gmap_vpids, cand_vpids = [], []
    visited = np.zeros((B, G), bool)
    step_ids = np.zeros((B, G), np.int64)
    for i in range(B):
        nv, gl, nc = min(int(nvis[i]), max(int(glen[i]) - 2, 0)), int(glen[i]), int(ncand[i])
        names = [None, None] + ['vp%d_%d' % (i, k) for k in range(gl - 2)]
        gmap_vpids.append(names)
        visited[i, 2:2 + nv] = True
        step_ids[i, 2:2 + nv] = np.arange(1, nv + 1)
        unv = names[2 + nv:]
        # candidates: (maybe) the previously visited node + unvisited nodes, in a hashed order
        order = np.argsort(uniform('cand_order%d' % i, (len(unv),), seed))
        cands = [unv[j] for j in order[:nc]]
        if nv > 1 and nc > 1 and cands:             # small graphs: fewer unvisited nodes than candidates
            cands[int(integers('bwpos%d' % i, (), 0, nc, seed)) % len(cands)] = names[2 + nv - 2]
        cand_vpids.append([None, None] + cands)
    gm = np.arange(G)[None] < glen[:, None]
    gmap_masks = gm.copy(); gmap_masks[:, 1] = False
    img = np.where(gm[..., None], uniform('gmap_img', (B, G, 768), seed), 0.0).astype(np.float32)
    img[:, 0] = 0.0
    pos = np.where(gm[..., None], uniform('gmap_pos', (B, G, 7), seed), 0.0).astype(np.float32)
    d = uniform('gmap_dist', (B, G, G), seed, 0.0, 10.0)
    d = np.triu(d, 1); d = d + d.transpose(0, 2, 1)
    real = gm.copy(); real[:, :2] = False
    d = np.where(real[:, :, None] & real[:, None, :], d, 0.0).astype(np.float32)
    vp_masks = np.arange(Vp)[None] < vp_lens[:, None]
    vp_img = uniform('vp_img', (B, Vp, 768), seed); vp_img[:, 0] = 0.0
    vp_img = np.where(vp_masks[..., None], vp_img, 0.0).astype(np.float32)
    vp_pos = np.zeros((B, Vp, 14), np.float32)
    vp_pos[:, :, :7] = uniform('vp_pos_start', (B, 1, 7), seed)
    cpos = uniform('vp_pos_cand', (B, Vp, 7), seed)
    cm = (np.arange(Vp)[None] >= 2) & (np.arange(Vp)[None] < 2 + ncand[:, None])
    vp_pos[:, :, 7:] = np.where(cm[..., None], cpos, 0.0)
    ones, zeros = np.ones((B, 1), bool), np.zeros((B, 1), bool)
    return {
        'gmap_img_embeds': img, 'gmap_step_ids': step_ids, 'gmap_pos_fts': pos,
        'gmap_masks': gmap_masks, 'gmap_pair_dists': d, 'gmap_visited_masks': visited,
        'gmap_vpids': gmap_vpids, 'gmap_lens': glen,
        'vp_img_embeds': vp_img, 'vp_pos_fts': vp_pos, 'vp_masks': vp_masks,
        'vp_nav_masks': np.concatenate([ones, zeros, nav_types == 1], 1),
        'vp_obj_masks': np.concatenate([zeros, zeros, nav_types == 2], 1) if reverie else None,
        'vp_cand_vpids': cand_vpids,
    }


# --------------------------------------------------------------------------
# pre-training trajectory batches (pretrain_src/data/tasks.py collators, dataset.py get_input)
# --------------------------------------------------------------------------
def pretrain_batch(B, L=40, O=3, seed=0, task='sap', min_steps=2, max_steps=4, V=36, n_classes=1000, reverie=False, No=20):
    """One collated batch for GlocalTextPathCMTPreTraining.forward(batch, task).
    Trajectories of T_i steps; graph-map tokens [stop], visited (= trajectory viewpoints), unvisited (every
    candidate ever seen that was never visited; some are seen from two steps, so the aggregation averages)."""
    lb = language_batch(B, L, seed, min_len=min(12, L))
    steps = integers('pt_steps', (B,), min_steps, max_steps + 1, seed)
    S = int(steps.sum())
    ncand_all = integers('pt_ncand', (S,), 3, 7, seed)
    traj_vpids, traj_cand_vpids, gmap_vpids, gmap_visited, gmap_steps = [], [], [], [], []
    s0 = 0
    for i in range(B):
        T = int(steps[i])
        path = ['t%d_v%d' % (i, t) for t in range(T)]
        cands_i, unv = [], []
        for t in range(T):
            nc = int(ncand_all[s0 + t])
            c = []
            if t + 1 < T:
                c.append(path[t + 1])
            if t > 0:
                c.append(path[t - 1])
                c.append('t%d_u%d_0' % (i, t - 1))        # seen again from this step -> averaged
            k = 0
            while len(c) < nc:
                c.append('t%d_u%d_%d' % (i, t, k)); k += 1
            order = np.argsort(uniform('pt_corder%d_%d' % (i, t), (len(c),), seed))
            c = [c[j] for j in order]
            cands_i.append(c)
            for x in c:
                if x not in path and x not in unv:
                    unv.append(x)
        s0 += T
        traj_vpids.append(path)
        traj_cand_vpids.append(cands_i)
        gmap_vpids.append([None] + path + unv)
        gmap_visited.append([0] + [1] * T + [0] * len(unv))
        gmap_steps.append([0] + list(range(1, T + 1)) + [0] * len(unv))
    glens = np.array([len(x) for x in gmap_vpids], np.int64)
    G = int(glens.max())
    pad = lambda rows, fill, dt: np.array([r + [fill] * (G - len(r)) for r in rows], dt)
    gm = np.arange(G)[None] < glens[:, None]
    d = uniform('pt_dist', (B, G, G), seed, 0.0, 10.0)
    d = np.triu(d, 1); d = d + d.transpose(0, 2, 1)
    real = gm.copy(); real[:, 0] = False
    d = np.where(real[:, :, None] & real[:, None, :], d, 0.0).astype(np.float32)
    # flattened per-step panoramas
    nav_types = (np.arange(V)[None] < np.array([len(c) for cs in traj_cand_vpids for c in cs])[:, None]).astype(np.int64)
    objv2, objv2_lens = _objv2('pt_objv2', S, V, O, seed)
    loc = np.concatenate([_angle_feats('pt_loc', (S, V), seed), np.ones((S, V, 3), np.float32)], -1)
    vp_pos = np.zeros((B, V + 1, 14), np.float32)
    vp_pos[:, :, :7] = uniform('pt_vp_start', (B, 1, 7), seed)
    last_nc = np.array([len(cs[-1]) for cs in traj_cand_vpids])
    cm = (np.arange(V + 1)[None] >= 1) & (np.arange(V + 1)[None] < 1 + last_nc[:, None])
    vp_pos[:, :, 7:] = np.where(cm[..., None], uniform('pt_vp_cand', (B, V + 1, 7), seed), 0.0)
    batch = {
        'txt_ids': lb['txt_ids'], 'txt_lens': lb['txt_lens'],
        'traj_view_img_fts': uniform('pt_view', (S, V, 768), seed), 'traj_obj_img_fts': None,
        'traj_loc_fts': loc, 'traj_nav_types': nav_types, 'traj_step_lens': [int(x) for x in steps],
        'traj_vp_view_lens': np.full((S,), V, np.int64), 'traj_vp_obj_lens': None,
        'traj_vpids': traj_vpids, 'traj_cand_vpids': traj_cand_vpids,
        'traj_objv2_fts': objv2, 'traj_objv2_lens': objv2_lens,
        'gmap_lens': glens, 'gmap_step_ids': pad(gmap_steps, 0, np.int64),
        'gmap_pos_fts': np.where(gm[..., None], uniform('pt_gpos', (B, G, 7), seed), 0.0).astype(np.float32),
        'gmap_pair_dists': d, 'gmap_vpids': gmap_vpids, 'vp_pos_fts': vp_pos,
        'gmap_visited_masks': pad(gmap_visited, 0, np.int64).astype(bool),
        'act_txt_ids': None, 'obj_txt_ids': None, 'act_lens': None, 'obj_lens': None,
    }
    if task.startswith('mlm'):
        lab = np.full(lb['txt_ids'].shape, -1, np.int64)
        pick = (uniform('pt_mlm_pick', lab.shape, seed, 0.0, 1.0) < 0.15) & lb['txt_masks']
        pick[:, 0] = False
        pick[:, 1] = True                                   # at least one masked token per row
        lab[pick] = lb['txt_ids'][pick]
        ids = lb['txt_ids'].copy(); ids[pick] = 103
        batch['txt_ids'], batch['txt_labels'] = ids, lab
    elif task.startswith('mrc'):
        m = uniform('pt_mrc_pick', (B, V), seed, 0.0, 1.0) < 0.15
        m[:, 0] = True
        p = uniform('pt_mrc_probs', (B, V, n_classes), seed, 0.0, 1.0) ** 4
        batch['vp_view_mrc_masks'] = m
        batch['vp_view_probs'] = (p / p.sum(-1, keepdims=True)).astype(np.float32)
        batch['vp_obj_mrc_masks'], batch['vp_obj_probs'] = None, None
    if task.startswith('sap') or task.startswith('valid_sap_og'):
        # teacher action = a hashed unvisited candidate of the last step (global index) / its local slot, or stop
        ga, la = np.zeros((B,), np.int64), np.zeros((B,), np.int64)
        pickc = integers('pt_sap_pick', (B,), 0, 8, seed)
        for i in range(B):
            last = traj_cand_vpids[i][-1]
            opts = [(gmap_vpids[i].index(c), j + 1) for j, c in enumerate(last) if c not in traj_vpids[i]]
            if pickc[i] < 6 and opts:
                ga[i], la[i] = opts[int(pickc[i]) % len(opts)]
        batch['global_act_labels'], batch['local_act_labels'] = ga, la
    if reverie:
        _add_objv1(batch, B, S, V, No, steps, seed, task, n_classes)
    return batch


def _add_objv1(batch, B, S, V, No, steps, seed, task, n_classes):
    """REVERIE pre-training layout (pretrain_src/data/dataset.py get_input + tasks.py collators): every step carries
    0..No 768-d "objv1" object tokens behind its 36 views; loc / nav-type rows are [views ; objects ; pad] per step,
    padded over all steps; the viewpoint tokens are [stop] + the LAST step's views and objects."""
    ol = integers('pt_objv1_lens', (S,), 0, No + 1, seed)
    last = np.cumsum(steps) - 1
    ol[0] = No                                   # pad_tensors pads to the longest: keep shapes a function of No
    free = [i for i in range(1, S) if i not in last]
    if free:
        ol[free[0]] = 0                          # an object-free step (the obj_len == 0 branch, vilmodel_reverie.py:559)
    ol[last] = np.maximum(ol[last], 1)           # the last step of every trajectory has >= 1 object (an OG target exists)
    Nm = int(ol.max())
    ov = np.arange(Nm)[None] < ol[:, None]
    fts = np.where(ov[..., None], uniform('pt_objv1_fts', (S, Nm, 768), seed), 0.0).astype(np.float32)
    oloc = np.concatenate([_angle_feats('pt_oloc', (S, Nm), seed), uniform('pt_obox', (S, Nm, 3), seed, 0.0, 1.0)], -1)
    oloc = np.where(ov[..., None], oloc, 0.0).astype(np.float32)
    batch['traj_obj_img_fts'], batch['traj_vp_obj_lens'] = fts, ol.astype(np.int64)
    batch['traj_loc_fts'] = np.concatenate([batch['traj_loc_fts'], oloc], 1)
    batch['traj_nav_types'] = np.concatenate([batch['traj_nav_types'], np.where(ov, 2, 0)], 1).astype(np.int64)
    lo = ol[last]
    T = V + int(lo.max()) + 1
    vp = np.zeros((B, T, 14), np.float32)
    vp[:, :V + 1] = batch['vp_pos_fts']
    vp[:, :, :7] = batch['vp_pos_fts'][:, :1, :7]
    valid = np.arange(T)[None] < (V + 1 + lo)[:, None]
    batch['vp_pos_fts'] = np.where(valid[..., None], vp, 0.0).astype(np.float32)
    if task.startswith('mrc'):
        Lm = int(lo.max())
        m = (uniform('pt_mrc_opick', (B, Lm), seed, 0.0, 1.0) < 0.3) & (np.arange(Lm)[None] < lo[:, None])
        m[:, 0] = True
        p = uniform('pt_mrc_oprobs', (B, Lm, n_classes), seed, 0.0, 1.0) ** 4
        batch['vp_obj_mrc_masks'] = m
        batch['vp_obj_probs'] = (p / p.sum(-1, keepdims=True)).astype(np.float32)
    if task.startswith('og') or task.startswith('valid_sap_og'):
        batch['obj_labels'] = (integers('pt_og_label', (B,), 0, 1 << 20, seed) % lo).astype(np.int64)


# --------------------------------------------------------------------------
# rollouts on a synthetic connectivity graph (what the simulator's observations look like to the graph-map code,
# finetune_src/r2r/env.py:_get_obs -> agent.py:406-408, 565-569)
# --------------------------------------------------------------------------
def episode_walks(B, T, seed=0, world=40, max_cand=6):
    """B episodes of T observations.  Returns obs[t][i] = {'viewpoint', 'position' (x,y,z), 'heading', 'elevation',
    'candidate': [{'viewpointId', 'position'}]} and ended[t][i] (an episode keeps returning its last observation once ended,
    as the environment does).  Worlds are random point sets joined to their nearest neighbours; the walk prefers
    unvisited neighbours, so graphs grow by 2-6 nodes a step and some nodes are seen from several places."""
    obs, ended = [[None] * B for _ in range(T)], np.zeros((T, B), bool)
    for i in range(B):
        pos = uniform('walk_pos%d' % i, (world, 3), seed, -12.0, 12.0).astype(np.float64)
        pos[:, 2] = uniform('walk_z%d' % i, (world,), seed, -1.5, 1.5)
        d = np.sqrt(((pos[:, None] - pos[None]) ** 2).sum(-1))
        deg = integers('walk_deg%d' % i, (world,), 2, max_cand + 1, seed)
        nbr = [set(np.argsort(d[v])[1:1 + int(deg[v])].tolist()) for v in range(world)]
        for v in range(world):                      # make it symmetric, keep the cap
            for u in list(nbr[v]):
                if len(nbr[u]) < max_cand:
                    nbr[u].add(v)
        nbr = [[u for u in sorted(nbr[v]) if v in nbr[u]] or [int(np.argsort(d[v])[1])] for v in range(world)]
        pick = uniform('walk_pick%d' % i, (T,), seed, 0.0, 1.0)
        head = uniform('walk_head%d' % i, (T,), seed, 0.0, 6.2831853).astype(np.float64)
        elev = uniform('walk_elev%d' % i, (T,), seed, -0.5, 0.5).astype(np.float64)
        stop_at = int(integers('walk_stop%d' % i, (1,), max(2, T // 2), T + 3, seed)[0])
        cur, seen, name = int(integers('walk_start%d' % i, (1,), 0, world, seed)[0]), set(), lambda v: 'e%d_%02d' % (i, v)
        for t in range(T):
            if t >= stop_at:
                ended[t, i] = True
                obs[t][i] = obs[t - 1][i]
                continue
            seen.add(cur)
            order = np.argsort(uniform('walk_ord%d_%d' % (i, t), (len(nbr[cur]),), seed))
            cands = [nbr[cur][j] for j in order]
            obs[t][i] = {'viewpoint': name(cur), 'position': tuple(float(x) for x in pos[cur]), 'heading': float(head[t]),
                         'elevation': float(elev[t]),
                         'candidate': [{'viewpointId': name(u), 'position': tuple(float(x) for x in pos[u])} for u in cands]}
            fresh = [u for u in cands if u not in seen]
            pool = fresh if (fresh and pick[t] < 0.8) else cands
            cur = pool[int(pick[t] * 1e6) % len(pool)]
    return obs, ended
