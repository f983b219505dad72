"""HIP-graph replay of one nav-step (panorama + navigation) for serving loops.

One nav-step is ~150 kernel launches of 2-30 us each; launched eagerly from Python the step is host-bound.
`GraphedNavStep` captures the step once into a HIP graph over static input/output buffers and replays it:
the caller copies the step's inputs into `.pano` / `.nav` buffers (or writes them in place), calls
`set_text()` once per episode (recomputes the hoisted text K/V panel) and `step()` per time step.
The agent-side glue that the reference does between the two calls (agent.py:231-237: [stop],[MEM] + panorama
tokens -> vp_img_embeds; current node's graph-map embedding = GAA-pooled panorama, agent.py:443-447) is part
of the captured step, and the RMF memory (cls_embeds) is fed back into the [MEM] slots for the next step."""
import torch


class GraphedNavStep:
    def __init__(self, model, txt_embeds, txt_masks, pano_batch, nav_batch, use_graph=True, reverie=False):
        self.model, self.reverie = model, reverie
        dev = txt_embeds.device
        self.txt = txt_embeds.clone()
        self.txt_masks = txt_masks.clone()
        self.pano = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in pano_batch.items()}
        self.nav = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in nav_batch.items()}
        B, G = self.nav['gmap_masks'].shape
        Vp = self.nav['vp_masks'].shape[1]
        if 'gmap_src' not in self.nav:
            src, bw = model.fusion_indices(self.nav['gmap_vpids'], self.nav['gmap_visited_masks'], self.nav['vp_cand_vpids'], G, Vp)
            self.nav['gmap_src'], self.nav['bw_mask'] = torch.from_numpy(src).to(dev), torch.from_numpy(bw).to(dev)
        H, cd = model.config.hidden_size, model.compute_dtype
        # static compute-dtype token buffers of the navigation call: slot 1 = [MEM] (fed back from cls_embeds), graph-map
        # slot 2 = the current node (GAA-pooled panorama), viewpoint slots 2.. = this step's panorama tokens
        self.gmap_img = self.nav['gmap_img_embeds'].to(cd).clone()
        self.vp_img = torch.zeros(B, Vp, H, device=dev, dtype=cd)
        self.gmap_img[:, 1].zero_()
        with torch.no_grad():
            self.kv = model.text_kv(self.txt)
        self.out = None
        self.graph = None
        if use_graph:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s), torch.no_grad():
                for _ in range(2):
                    self._body()
            torch.cuda.current_stream().wait_stream(s)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph), torch.no_grad():
                self._body()

    def set_text(self, txt_embeds, txt_masks):
        """New episode batch: copy the instruction encoding in, recompute the text K/V panel, clear [MEM]."""
        with torch.no_grad():
            self.txt.copy_(txt_embeds)
            self.txt_masks.copy_(txt_masks)
            self.model.text_kv(self.txt, out=self.kv)
            self.gmap_img[:, 1].zero_()
            self.vp_img[:, 1].zero_()

    def set_gmap_img(self, gmap_img_embeds):
        """New graph-map node embeddings (any float dtype); slots 1 and 2 are owned by the runner."""
        with torch.no_grad():
            mem = self.vp_img[:, 1].clone()
            self.gmap_img.copy_(gmap_img_embeds)
            self.gmap_img[:, 1] = mem

    def _body(self):
        m = self.model
        pe, pm, pf = m('panorama', self.pano)
        nb = dict(self.nav)
        self.vp_img[:, 2:] = pe
        self.gmap_img[:, 2] = pf
        nb.update(vp_img_embeds=self.vp_img, gmap_img_embeds=self.gmap_img,
                  txt_embeds=self.txt, txt_masks=self.txt_masks, txt_kv=self.kv)
        o = m('navigation', nb)
        self.vp_img[:, 1] = o['cls_embeds']
        self.gmap_img[:, 1] = o['cls_embeds']
        self.out = {'pano_embeds': pe, 'pano_fused': pf, **o}

    def step(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            with torch.no_grad():
                self._body()
        return self.out
