"""HIP-graph replay of one nav-step (panorama + navigation) for serving loops.

One nav-step is ~150 kernel launches of 2-30 us each; launched eagerly from Python the step is host-bound.
`GraphedNavStep` captures the step once into a HIP graph over static input/output buffers and replays it:
the caller copies the step's inputs into `.pano` / `.nav` buffers (or writes them in place), calls
`set_text()` once per episode (recomputes the hoisted text K/V panel) and `step()` per time step.
The agent-side glue that the reference does between the two calls (agent.py:231-237: [stop],[MEM] + panorama
tokens -> vp_img_embeds; current node's graph-map embedding = GAA-pooled panorama, agent.py:443-447) is part
of the captured step, and the RMF memory (cls_embeds) is fed back into the [MEM] slots for the next step.

Graph-map token order is [stop], [MEM], visited nodes in insertion order, unvisited nodes (agent.py:165-187), so the
CURRENT node's token is slot 2 only on the first step of an episode; `cur_gmap_idx` (int64 [B], a static buffer the
caller updates with `set_current_node`, e.g. from BatchedGraphMap's slot table) says where the pooled panorama goes.

`split=True` captures the two calls as two graphs (`pano_step()`, `nav_step()`) so that a device-resident graph map
(dsrg_amd.graphmap.BatchedGraphMap: update_embeds -> nav_inputs, agent.py:443-468) can run between them on this
step's panorama; `load_nav()` pads its output into the static buffers.  That is the exact rollout protocol; the
single-graph `step()` is the serving shortcut in which only the current node's embedding changes inside the step.

The captured graph holds raw addresses of the model's packed weights.  The runner keeps those tensors alive and
re-captures when the model re-packs them (eval() / train() / .to() / load_state_dict() / an optimizer step)."""
import torch

from ._lib import DsrgHipError
from . import functional as Fn


class GraphedNavStep:
    NAV_KEYS = ('gmap_step_ids', 'gmap_pos_fts', 'gmap_masks', 'gmap_pair_dists', 'gmap_visited_masks', 'vp_pos_fts', 'vp_masks',
                'vp_nav_masks', 'vp_obj_masks', 'gmap_src', 'bw_mask')

    def __init__(self, model, txt_embeds, txt_masks, pano_batch, nav_batch, use_graph=True, reverie=False, split=False):
        self.model, self.reverie, self.split = model, reverie, split
        dev = txt_embeds.device
        self.txt = txt_embeds.clone()
        self.txt_masks = txt_masks.clone()
        self.pano = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in pano_batch.items()}
        # the 768-d image features are held in the compute dtype: the cast is paid once when a step's inputs are loaded
        # (`runner.pano['view_img_fts'].copy_(fts)` converts), not by a kernel inside every replayed step
        for k in ('view_img_fts',) + (('obj_img_fts',) if reverie else ()):
            if isinstance(self.pano.get(k), torch.Tensor) and self.pano[k].is_floating_point():
                self.pano[k] = self.pano[k].to(model.compute_dtype)
        self.nav = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in nav_batch.items()}
        B, G = self.nav['gmap_masks'].shape
        Vp = self.nav['vp_masks'].shape[1]
        if 'gmap_src' not in self.nav:
            src, bw = model.fusion_indices(self.nav['gmap_vpids'], self.nav['gmap_visited_masks'], self.nav['vp_cand_vpids'], G, Vp)
            self.nav['gmap_src'], self.nav['bw_mask'] = torch.from_numpy(src).to(dev), torch.from_numpy(bw).to(dev)
        H, cd = model.config.hidden_size, model.compute_dtype
        # static compute-dtype token buffers of the navigation call: slot 1 = [MEM] (fed back from cls_embeds), graph-map
        # slot cur_gmap_idx = the current node (GAA-pooled panorama), viewpoint slots 2.. = this step's panorama tokens
        self.gmap_img = self.nav['gmap_img_embeds'].to(cd).clone()
        self.vp_img = torch.zeros(B, Vp, H, device=dev, dtype=cd)
        self.gmap_img[:, 1].zero_()
        self.cur_gmap_idx = torch.full((B,), 2, device=dev, dtype=torch.int64)
        self._rows = torch.arange(B, device=dev)
        self.use_graph = use_graph
        self.out, self.pano_out = None, None
        self.graph = self.graph_pano = self.graph_nav = None
        self._capture()

    def _capture(self):
        """(Re)build everything that bakes in the packed weights' addresses: the text K/V panel and the HIP graph(s)."""
        model = self.model
        with torch.no_grad():
            self._W = model._weights()            # keep the pack alive: the graphs replay raw pointers into it
            self._token = model._wplan_token()
            self._wver = self._pack_version()
            self.kv = model.text_kv(self.txt) if getattr(self, 'kv', None) is None else model.text_kv(self.txt, out=self.kv)
        self.graph = self.graph_pano = self.graph_nav = None
        if not self.use_graph:
            return
        # capture leaves the recurrent state ([MEM] slots) as it found it: warm-up and capture run the body for real
        keep = (self.vp_img[:, 1].clone(), self.gmap_img.clone())

        def cap(fn):
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s), torch.no_grad():
                for _ in range(2):
                    fn()
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g), torch.no_grad():
                fn()
            return g
        if self.split:
            self.graph_pano = cap(self._pano_body)
            self.graph_nav = cap(self._nav_body)
        else:
            self.graph = cap(self._body)
        with torch.no_grad():
            self.vp_img[:, 1] = keep[0]
            self.gmap_img.copy_(keep[1])

    def _pack_version(self):
        plan = self.model._wplan
        return None if plan is None else Fn.pack_version(plan[1].srcs)

    def _check_weights(self):
        """The model dropped or re-packed its weight cache since capture -> capture again (cheap next to serving stale
        or freed weights silently)."""
        m = self.model
        # eval(): the model itself treats its pack as frozen until train()/eval()/.to()/load_state_dict() (all of which
        # change the token); in train() mode an optimizer step may have moved the parameters: compare their versions
        if m._wplan_token() is not self._token or m._wcache is not self._W or (m.training and self._pack_version() != self._wver):
            if torch.cuda.is_current_stream_capturing():
                raise DsrgHipError('GraphedNavStep: model weights changed while a capture is in progress')
            self._capture()

    def set_text(self, txt_embeds, txt_masks):
        """New episode batch: copy the instruction encoding in, recompute the text K/V panel, clear [MEM]."""
        self._check_weights()
        with torch.no_grad():
            self.txt.copy_(txt_embeds)
            self.txt_masks.copy_(txt_masks)
            self.model.text_kv(self.txt, out=self.kv)
            self.gmap_img[:, 1].zero_()
            self.vp_img[:, 1].zero_()
            self.cur_gmap_idx.fill_(2)

    def set_current_node(self, cur_gmap_idx):
        """Position of each episode's current node in the graph-map token order (int tensor / list [B], >= 2)."""
        with torch.no_grad():
            self.cur_gmap_idx.copy_(torch.as_tensor(cur_gmap_idx, device=self.cur_gmap_idx.device, dtype=torch.int64))

    def set_gmap_img(self, gmap_img_embeds):
        """New graph-map node embeddings (any float dtype); slot 1 ([MEM]) and the current node's slot (cur_gmap_idx,
        rewritten by the next step) are owned by the runner."""
        with torch.no_grad():
            mem = self.vp_img[:, 1].clone()
            self.gmap_img.copy_(gmap_img_embeds)
            self.gmap_img[:, 1] = mem

    def load_nav(self, nav_inputs):
        """Copy a step's navigation inputs (e.g. BatchedGraphMap.nav_inputs(), G <= the static G) into the static buffers,
        padding the graph-map side behind the masks: padded nodes are masked, unvisited, distance 0, no fusion source.
        Graph-map slot 1 keeps the runner's [MEM]."""
        G = self.gmap_img.shape[1]
        with torch.no_grad():
            g = nav_inputs['gmap_masks'].shape[1]
            if g > G or nav_inputs['vp_masks'].shape[1] != self.vp_img.shape[1]:
                raise DsrgHipError('GraphedNavStep.load_nav: %d graph nodes > static %d, or viewpoint length differs' % (g, G))
            for k in self.NAV_KEYS:
                src, dst = nav_inputs.get(k), self.nav.get(k)
                if src is None or dst is None:
                    if (src is None) != (dst is None):
                        raise DsrgHipError('GraphedNavStep.load_nav: %s present on one side only' % k)
                    continue
                if k.startswith('gmap') and g < G:
                    dst.fill_(-1 if k == 'gmap_src' else 0)
                if k == 'gmap_pair_dists':
                    dst[:, :g, :g] = src
                elif k.startswith('gmap'):
                    dst[:, :g] = src.to(dst.dtype)
                else:
                    dst.copy_(src)
            mem = self.vp_img[:, 1].clone()
            if g < G:
                self.gmap_img.zero_()
            self.gmap_img[:, :g] = nav_inputs['gmap_img_embeds'].to(self.gmap_img.dtype)
            self.gmap_img[:, 1] = mem

    # No copy kernels between the calls: the panorama call's last LayerNorm stores its tokens straight into slots 2.. of the
    # viewpoint-token buffer (`pano_out`), the GAA kernel also stores the pooled embedding into the current node's graph-map
    # slot (`gmap_scatter`), the RMF LayerNorm also stores the memory into both [MEM] slots (`mem_out`), and the attention
    # kernels form the length masks themselves (`want_pano_masks=False`; pano_masks() builds the tensor on demand).
    def _pano_body(self, scatter=False):
        pb = dict(self.pano)
        pb.update(pano_out=self.vp_img[:, 2:], want_pano_masks=False)
        if scatter:
            pb['gmap_scatter'] = (self.gmap_img, self.cur_gmap_idx)
        pe, pm, pf = self.model('panorama', pb)
        self.pano_out = (pe, pm, pf)

    def pano_masks(self):
        """bool [B, T] validity of the panorama tokens of the current inputs (the panorama call's second return value)"""
        lens = self.pano['view_lens']
        if self.reverie and self.pano.get('obj_lens') is not None and self.pano.get('obj_img_fts') is not None:
            lens = lens + self.pano['obj_lens']
        T = self.vp_img.shape[1] - 2
        return torch.arange(T, device=lens.device)[None] < lens[:, None]

    def _nav_body(self):
        nb = dict(self.nav)
        nb.update(vp_img_embeds=self.vp_img, gmap_img_embeds=self.gmap_img,
                  txt_embeds=self.txt, txt_masks=self.txt_masks, txt_kv=self.kv,
                  mem_out=(self.vp_img[:, 1], self.gmap_img[:, 1]))
        o = self.model('navigation', nb)
        pe, pm, pf = self.pano_out
        self.out = {'pano_embeds': pe, 'pano_fused': pf, **o}

    def _body(self):
        self._pano_body(scatter=True)
        self._nav_body()

    def _run(self, graph, fn):
        self._check_weights()
        if graph is not None:
            graph.replay()
        else:
            with torch.no_grad():
                fn()

    def step(self):
        """panorama + navigation in one replay; the pooled panorama goes to the current node's graph-map slot."""
        if self.split:
            raise DsrgHipError('GraphedNavStep(split=True): call pano_step() / nav_step()')
        self._run(self.graph, self._body)
        return self.out

    def pano_step(self):
        """split mode: the panorama call -> (pano_embeds, pano_masks, pano_fused), static tensors"""
        if not self.split:
            raise DsrgHipError('GraphedNavStep: pano_step() needs split=True')
        self._run(self.graph_pano, self._pano_body)
        return self.pano_out

    def nav_step(self):
        """split mode: the navigation call on the static buffers (fill them with load_nav() / set_gmap_img())"""
        if not self.split:
            raise DsrgHipError('GraphedNavStep: nav_step() needs split=True')
        self._run(self.graph_nav, self._nav_body)
        return self.out
