"""One optimisation step of the pre-training path as a replayed HIP graph.

The reference's step (pretrain_src/train_r2r.py:239-295) is `forward(batch, task) -> loss.mean().backward() ->
clip_grad_norm_ -> optimizer.step()`.  On an MI355X it is ~900 kernels of 5-30 us each, so eager Python (autograd node
walk, ~1800 ctypes calls, ~600 gradient allocations) costs as much wall time as the kernels do.  `GraphedTrainStep`
captures the whole step once per (task, batch shape signature) and replays it:

  * everything the forward builds on the HOST from the batch's index data (aggregation incidences, selections of masked
    tokens, last-step rows, logit-fusion sources) is built before the capture (`model.prepare`, kept under
    batch['_prep']), so the captured region contains kernels only;
  * dropout masks are a function of (seed, element index); the seeds are captured by value, and every dropout kernel
    adds a device word that the graph advances once per replay (dsrg_set_dropout_offset), so each replay draws new
    masks while its forward and backward still agree;
  * the cast / packed copies of the parameters are produced inside the graph, i.e. re-made from the updated parameters
    on every replay; functional.PACK_EPOCH moves after each replay so that eager callers never reuse a captured pack;
  * with more than one rank the step is two graphs around one eager all-reduce of ONE flat gradient buffer (bf16 on
    request): graph 1 = forward + backward + flatten, all-reduce, graph 2 = un-flatten (the parameters' .grad are views
    of the flat buffer) + clip + optimizer.  No DistributedDataParallel wrapper is involved, and every rank must run
    the same task (the reference broadcasts the task id, pretrain_src/data/loader.py:47-51).

A new signature costs one eager step (which IS that call's step) plus one capture; a production loader pads batches to
a few bucket shapes.  A hit copies the batch's tensors into the captured ones (skipped when the caller hands back the
captured batch itself, `static_batch`).
"""
import torch
import torch.distributed as dist

from . import functional as Fn
from . import kernels as K


def _tensors(obj, path=()):
    """(path, tensor) for every tensor reachable through dicts / lists / tuples of a batch, in a fixed order"""
    if isinstance(obj, torch.Tensor):
        yield path, obj
    elif isinstance(obj, dict):
        for k in sorted(obj, key=str):
            yield from _tensors(obj[k], path + (k,))
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            if isinstance(v, (torch.Tensor, dict, list, tuple)):
                yield from _tensors(v, path + (i,))


def _clone(obj):
    if isinstance(obj, torch.Tensor):
        return obj.clone()
    if isinstance(obj, dict):
        return {k: _clone(v) for k, v in obj.items()}
    if isinstance(obj, tuple):
        return tuple(_clone(v) for v in obj)
    if isinstance(obj, list):
        return [_clone(v) for v in obj]
    return obj


def signature(batch):
    return tuple((p, tuple(t.shape), t.dtype) for p, t in _tensors(batch))


class _Entry:
    __slots__ = ('static', 'tensors', 'g1', 'g2', 'loss', 'flat', 'grads', 'f32')


class GraphedTrainStep:
    def __init__(self, model, optimizer, max_norm, params=None, loss_of=None, bf16_grads=False, group=None, max_graphs=16, manage_packs=True):
        """model: GlocalTextPathCMTPreTraining (bare, not DDP-wrapped).  optimizer: constructed with capturable=True.
        loss_of(model, batch, task) -> scalar; default = the reference's `model(batch, task, compute_loss=True).mean()`."""
        self.model, self.opt, self.max_norm = model, optimizer, float(max_norm)
        self.params = list(params) if params is not None else [p for p in model.parameters() if p.requires_grad]
        self.loss_of = loss_of or (lambda m, b, t: m(b, t, compute_loss=True).mean())
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.group, self.bf16_grads = group, bool(bf16_grads)
        self.graphs, self.max_graphs = {}, max_graphs
        self.dev = self.params[0].device
        self.pool = torch.cuda.graph_pool_handle()
        # every step - eager, captured, replayed - runs on ONE side stream: autograd pins each parameter's AccumulateGrad node to
        # the stream it was created on, and a node left over from work on the default stream would pull that stream into the
        # capture (unjoined work at capture end)
        self.stream = torch.cuda.Stream(self.dev)
        self.seed_off = torch.zeros(1, dtype=torch.int64, device=self.dev)
        K.set_dropout_offset(self.seed_off)
        self.native_opt = hasattr(optimizer, 'clip_grad_norm') and hasattr(optimizer, 'sync_hyper')
        self.manage_packs = manage_packs and __import__('os').environ.get('DSRG_MANAGE_PACKS', '1') == '1'      # (env: A/B)
        self._attached = False
        for g in optimizer.param_groups:
            if not self.native_opt and not g.get('capturable', False):
                raise ValueError('GraphedTrainStep: construct a torch optimizer with capturable=True (or use dsrg_amd.optim.AdamW)')

    # ------------------------------------------------------------------ pieces of a step
    def _fwd_bwd(self, batch, task):
        loss = self.loss_of(self.model, batch, task)
        loss.backward()
        return loss.detach()

    def _update(self, grad_scale=1.0):
        if self.native_opt:                      # dsrg_amd.optim.AdamW: norm, clip coefficient and update in three launches
            self.opt.grad_scale = grad_scale
            self.opt.clip_grad_norm(self.max_norm)
            self.opt.step()
            return
        if grad_scale != 1.0:
            torch._foreach_mul_([p.grad for p in self.params if p.grad is not None], grad_scale)
        torch.nn.utils.clip_grad_norm_(self.params, self.max_norm)
        self.opt.step()

    def _allreduce(self, flat):
        if dist.get_backend(self.group) == 'gloo' and flat.is_cuda:        # rehearsal backend: staged through host memory
            cpu = flat.to('cpu', torch.float32)
            dist.all_reduce(cpu, group=self.group)
            flat.copy_(cpu.to(flat.dtype))
        else:
            dist.all_reduce(flat, group=self.group)

    def _eager(self, batch, task):
        """the step as the reference runs it (also the warm-up that fills batch['_prep'] and every lazily created buffer)"""
        self.opt.zero_grad(set_to_none=True)
        loss = self._fwd_bwd(batch, task)
        if self.world > 1:
            grads = [p.grad for p in self.params if p.grad is not None]
            flat = torch.cat([g.reshape(-1) for g in grads])
            flat = flat.to(torch.bfloat16) if self.bf16_grads else flat
            self._allreduce(flat)
            torch._foreach_copy_(grads, [f.view_as(g) for f, g in zip(flat.split([g.numel() for g in grads]), grads)])
        self._update(1.0 / self.world)
        return loss

    # ------------------------------------------------------------------ capture
    def _capture(self, batch, task):
        e = _Entry()
        e.static = _clone(batch)
        e.tensors = [t for _, t in _tensors(e.static)]
        side = self.stream
        self.opt.zero_grad(set_to_none=True)
        self._drop_packs()
        if self.native_opt:
            self.opt.reserve_capture()
        e.g1, e.g2, e.flat, e.f32 = torch.cuda.CUDAGraph(), None, None, None
        with torch.cuda.graph(e.g1, pool=self.pool, stream=side), K.zero_pool_capture_scope():
            try:
                self.seed_off.add_(1)
                e.loss = self._fwd_bwd(e.static, task)
                e.grads = [p.grad for p in self.params if p.grad is not None]
                if self.world > 1:
                    flat = torch.cat([g.reshape(-1) for g in e.grads])
                    e.flat = flat.to(torch.bfloat16) if self.bf16_grads else flat
                else:
                    self._update()
            except BaseException:
                # ending a capture that an error has invalidated can take the process down inside the HIP runtime before the
                # exception surfaces: say what happened first
                import sys, traceback
                print('GraphedTrainStep: error inside the stream capture of task %r:' % (task,), file=sys.stderr)
                traceback.print_exc()
                sys.stderr.flush()
                raise
        if self.world > 1:
            owners = [p for p in self.params if p.grad is not None]
            f32 = e.flat if e.flat.dtype == torch.float32 else torch.empty(e.flat.numel(), dtype=torch.float32, device=self.dev)
            e.f32 = f32            # graph 2 reads and writes it on every replay: it must outlive the parameters' .grad views
            for p, v in zip(owners, f32.split([g.numel() for g in e.grads])):
                p.grad = v.view_as(p)
            e.g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(e.g2, pool=self.pool, stream=side), K.zero_pool_capture_scope():
                if f32 is not e.flat:
                    f32.copy_(e.flat)
                self._update(1.0 / self.world)
        return e

    def _drop_packs(self):
        """differentiable weight packs cached by the model keep last step's autograd nodes alive; see self.stream"""
        drop = getattr(self.model, 'drop_grad_packs', None)
        if drop is not None:
            drop()

    # ------------------------------------------------------------------ API
    def static_batch(self, batch, task):
        """the captured copy of `batch` (captures on first use): hand it back to step() to skip the input copies"""
        if self._key(batch, task) not in self.graphs:
            self.step(batch, task)
        return self.graphs[self._key(batch, task)].static

    def _key(self, batch, task):
        if '_prep' not in batch:
            batch['_prep'] = {}
        prep = getattr(self.model, 'prepare', None)
        if prep is not None:
            prep(batch, task)
        return (task, signature(batch))

    def step(self, batch, task):
        """one optimisation step; returns the (device) loss of that step"""
        cur = torch.cuda.current_stream(self.dev)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            loss = self._step(batch, task)
        cur.wait_stream(self.stream)
        return loss

    def _step(self, batch, task):
        key = self._key(batch, task)
        e = self.graphs.get(key)
        if e is None:
            if not self.graphs:
                self._drop_packs()
            import warnings
            with warnings.catch_warnings(record=True) as seen:
                warnings.simplefilter('always')
                loss = self._eager(batch, task)
            for w in seen:
                if "AccumulateGrad node's stream does not match" in str(w.message):
                    # capturing now would pull that other stream into the capture and die in hipStreamEndCapture
                    raise RuntimeError('GraphedTrainStep: an autograd graph that was built on another stream is still alive (a loss or '
                                       'output tensor of an earlier forward of this model): delete it before the first step()')
                warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
            if self.native_opt and self.manage_packs and not self._attached and hasattr(self.model, 'weight_packs'):
                # from here on the optimizer's update kernel writes the bf16 weight copies: no cast pass in the captured forwards
                self.opt.attach_packs(self.model.weight_packs())
                self.opt.packed_grads = self.world == 1        # (N ranks: the flat all-reduce buffer is built from the .grad tensors)
                self.model.sync_packs()
                self._attached = True
            if len(self.graphs) >= self.max_graphs:
                self.graphs.pop(next(iter(self.graphs)))
            self.graphs[key] = self._capture(batch, task)
            return loss
        if batch is not e.static:
            torch._foreach_copy_(e.tensors, [t for _, t in _tensors(batch)])
        if self.native_opt:
            self.opt.sync_hyper()                # a schedule may have moved lr: device words, no recapture
        e.g1.replay()
        if e.g2 is not None:
            self._allreduce(e.flat)
            e.g2.replay()
        Fn.PACK_EPOCH[0] += 1
        return e.loss
