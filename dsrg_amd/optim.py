"""MI355X-native optimizer step of the pre-training path: the reference's `AdamW` (pretrain_src/optim/adamw.py:8-92, built
by optim/misc.py:9-33 with two weight-decay groups) and the `clip_grad_norm_` call in front of it
(pretrain_src/train_r2r.py:283-294), as three launches over ALL parameters (dsrg_grad_sumsq -> dsrg_adamw_prepare ->
dsrg_adamw_step, csrc/optim.hip) instead of ~10 multi-tensor launches + 6 passes over the gradients.

Same constructor and validation as the reference class; same arithmetic (moving averages, denom = sqrt(v) + eps,
bias-corrected step size, decoupled decay AFTER the update, per-parameter step counters, parameters without a gradient are
skipped).  Differences a caller can see:
  * clipping is `opt.clip_grad_norm(max_norm)` (returns the total norm as a device scalar); the coefficient is applied where
    the update reads the gradients, so `.grad` keeps the UNclipped values after the step (the reference never reads them);
  * state['step'] is a view into a device int32 counter tensor, not a Python int;
  * lr / weight_decay are read from device words: under a captured step (dsrg_amd.train_graph) a schedule that rewrites
    param_group['lr'] is picked up by `sync_hyper()` before each replay, without recapturing;
  * parameters and gradients must be fp32, contiguous, on one GPU; there is no CPU path.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .kernels import _p, _stream

_TENSOR = np.dtype([('p', '<u8'), ('g', '<u8'), ('m', '<u8'), ('v', '<u8'), ('numel', '<i8'), ('group', '<i4'), ('step_idx', '<i4'),
                    ('vec_ok', '<i4'), ('reserved_', '<i4'), ('g2', '<u8'), ('pack', '<u8')])          # = dsrg_opt_tensor_t (include/dsrg_hip.h), 72 bytes


class _Plan:
    __slots__ = ('key', 'host', 'dev', 'table', 'blocks', 'n_tensors', 'n_blocks', 'partials', 'consts')


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[1]))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(eps))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias))
        self._index = {}
        for g in self.param_groups:
            for p in g['params']:
                self._index[p] = len(self._index)
        self._dev = None
        self._plans = []                # most recent first; a plan is keyed by the (parameter, gradient) addresses it was built for
        self._captured = []
        self._pack_dst, self._pack_gen, self._pack_specs = {}, 0, []      # attach_packs()
        self.packed_grads = False       # attach_packs + True: bf16 gradients of packed weights are read as they lie (no fp32 .grad)
        self._gstash = {}               # parameter address -> (address of its bf16 gradient, the tensor that holds it); one backward
        self._pending = None            # plan whose prepare launch (clip_grad_norm) already ran for the coming step()
        self.grad_scale = 1.0           # multiplies every gradient where it is read (1 / world size after a summing all-reduce)

    # ------------------------------------------------------------------ device state
    PINNED_ARENA = 8 << 20          # bytes of pinned host memory per arena chunk: the tables of CAPTURED steps (one region each)

    def reserve_capture(self, nbytes=2 << 20):
        """Call OUTSIDE a stream capture before capturing a step: makes sure the pinned arena has `nbytes` left for that step's
        tables (pinned memory cannot be allocated inside a capture).  Older chunks stay alive with the plans that point into them."""
        self._ensure_device()
        if self._arena_used + nbytes > self._arena.numel():
            self._arena, self._arena_used = torch.empty(max(self.PINNED_ARENA, nbytes), dtype=torch.uint8).pin_memory(), 0

    def _ensure_device(self):
        if self._dev is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('dsrg_amd.optim.AdamW: run one eager step (or sync_hyper()) before capturing a step')
            self._init_device(next(iter(self._index)).device)

    def _init_device(self, dev):
        self._dev = dev
        if dev.type != 'cuda':
            raise _lib.DsrgHipError('dsrg_amd.optim.AdamW needs parameters on a GPU (got %s); there is no CPU path' % dev)
        self._arena, self._arena_used = torch.empty(self.PINNED_ARENA, dtype=torch.uint8).pin_memory(), 0
        self._steps = torch.zeros(max(1, len(self._index)), dtype=torch.int32, device=dev)
        self._norm = torch.zeros(2, dtype=torch.float32, device=dev)
        self._hp = torch.zeros(len(self.param_groups), 2, dtype=torch.float32, device=dev)
        self._hp_sent = None
        self._chunk = int(_lib.load().dsrg_opt_chunk())

    def sync_hyper(self):
        """upload (lr, weight_decay) of every group if they changed since the last upload; stream-ordered, never captured"""
        self._ensure_device()
        vals = tuple((float(g['lr']), float(g['weight_decay'])) for g in self.param_groups)
        if vals != self._hp_sent:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('dsrg_amd.optim.AdamW: lr / weight_decay changed inside a stream capture; call sync_hyper() before it')
            self._hp.copy_(torch.tensor(vals, dtype=torch.float32))
            self._hp_sent = vals

    def _plan(self):
        gst = self._gstash
        ents = [(gi, p) for gi, g in enumerate(self.param_groups) for p in g['params'] if p.grad is not None or (gst and p.data_ptr() in gst)]
        if not ents:
            return None
        self._ensure_device()
        capturing = torch.cuda.is_current_stream_capturing()
        key = (self._pack_gen,) + tuple((p.data_ptr(), 0 if p.grad is None else p.grad.data_ptr(), gst[p.data_ptr()][0] if p.data_ptr() in gst else 0)
                                        for _, p in ents)
        for pl in self._plans + self._captured:
            if pl.key == key:
                if pl.host is not None and not capturing:
                    pl.dev.copy_(pl.host, non_blocking=True)     # a captured plan's upload is a graph node: it may never have run
                return pl
        consts = None
        tab = np.zeros(len(ents), _TENSOR)
        for i, (gi, p) in enumerate(ents):
            g = p.grad
            g2 = gst[p.data_ptr()][0] if p.data_ptr() in gst else 0
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != self._dev or \
                    (g is not None and (g.dtype != torch.float32 or not g.is_contiguous())):
                raise _lib.DsrgHipError('dsrg_amd.optim.AdamW: parameters and gradients must be contiguous fp32 tensors on %s' % self._dev)
            grp = self.param_groups[gi]
            c = (float(grp['betas'][0]), float(grp['betas'][1]), float(grp['eps']), bool(grp['correct_bias']))
            if consts is not None and c != consts:
                raise _lib.DsrgHipError('dsrg_amd.optim.AdamW: betas / eps / correct_bias must be the same in every parameter group')
            consts = c
            st = self.state[p]
            if not st:
                if capturing:        # a zero fill inside the capture would wipe the moments on every replay
                    raise RuntimeError('dsrg_amd.optim.AdamW: a parameter gets its first gradient inside a stream capture; run one eager step first')
                st['exp_avg'], st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format), torch.zeros_like(p, memory_format=torch.contiguous_format)
                st['step'] = self._steps[self._index[p]]
            ptrs = (p.data_ptr(), 0 if g is None else g.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr())
            pack = self._pack_dst.get(p.data_ptr(), 0)
            tab[i] = ptrs + (p.numel(), gi, self._index[p], int(all(a % 16 == 0 for a in ptrs) and pack % 8 == 0 and g2 % 8 == 0), 0, g2, pack)
        nch = (tab['numel'] + self._chunk - 1) // self._chunk
        starts = np.cumsum(nch) - nch
        blocks = np.empty((int(nch.sum()), 2), np.int32)
        blocks[:, 0] = np.repeat(np.arange(len(ents), dtype=np.int32), nch)
        blocks[:, 1] = np.arange(blocks.shape[0], dtype=np.int64) - np.repeat(starts, nch)
        raw = np.concatenate([tab.view(np.uint8).reshape(-1), blocks.view(np.uint8).reshape(-1)])
        pl = _Plan()
        pl.key, pl.consts, pl.n_tensors, pl.n_blocks = key, consts, len(ents), blocks.shape[0]
        pl.dev = torch.empty(raw.shape[0], dtype=torch.uint8, device=self._dev)
        pl.table, pl.blocks = pl.dev.data_ptr(), pl.dev.data_ptr() + tab.nbytes           # 72-byte entries: the block table stays 8-byte aligned
        pl.partials = torch.empty(pl.n_blocks, dtype=torch.float32, device=self._dev)
        if capturing:
            # a memcpy node that re-reads its pinned source on every replay: the source is a region of the arena that is
            # never handed out again (pinned memory cannot be allocated inside a capture, hence the arena)
            n = (raw.shape[0] + 63) // 64 * 64
            if self._arena_used + n > self._arena.numel():
                raise RuntimeError('dsrg_amd.optim.AdamW: pinned table arena exhausted (%d captured plans): call reserve_capture() before the capture' % len(self._captured))
            pl.host = self._arena[self._arena_used:self._arena_used + raw.shape[0]]
            self._arena_used += n
            pl.host.copy_(torch.from_numpy(raw))
            pl.dev.copy_(pl.host, non_blocking=True)
            self._captured.append(pl)            # captured plans live as long as the optimizer: their graphs replay the copy
        else:
            pl.host = None
            pl.dev.copy_(torch.from_numpy(raw))      # pageable source: staged synchronously, stream-ordered
        self._plans.insert(0, pl)
        del self._plans[8:]
        return pl

    def _prepare(self, pl, max_norm):
        lib = _lib.load()
        n_part = 0
        if max_norm is not None and max_norm > 0:
            _lib.check(lib.dsrg_grad_sumsq(pl.table, pl.blocks, pl.n_blocks, _p(pl.partials), _stream()), 'dsrg_grad_sumsq')
            n_part = pl.n_blocks
        _lib.check(lib.dsrg_adamw_prepare(_p(pl.partials), n_part, float(max_norm or 0.0), float(self.grad_scale), _p(self._norm),
                                          pl.table, pl.n_tensors, _p(self._steps), _stream()), 'dsrg_adamw_prepare')
        self._pending = pl

    # ------------------------------------------------------------------ packed-weight hand-over
    def attach_packs(self, specs):
        """specs: the model's weight packs (functional._Spec objects that carry a plan, see functional.manage_pack).  From now
        on step() also writes every updated parameter, rounded to bf16, into its place in the packs' persistent buffers, and
        the packs' forward skips its cast pass over those parameters (0.4 ms per pre-training step).  Contract: while
        attached, parameters change only through this optimizer (load_state_dict / `.data =` are detected and re-synced)."""
        from . import functional as Fn
        self._pack_dst, self._pack_specs = {}, list(specs)
        for spec in self._pack_specs:
            for src_ptr, dst_ptr in Fn.manage_pack(spec, self):
                self._pack_dst.setdefault(src_ptr, dst_ptr)
        self._pack_gen += 1            # cached tables were built without (or with other) destinations

    def detach_packs(self):
        from . import functional as Fn
        for spec in self._pack_specs:
            Fn.manage_pack(spec, None)
        self._pack_dst, self._pack_specs = {}, []
        self._gstash.clear()
        self._pack_gen += 1

    # ------------------------------------------------------------------ API
    @torch.no_grad()
    def clip_grad_norm(self, max_norm):
        """torch.nn.utils.clip_grad_norm_(all parameters, max_norm) folded into the coming step(): returns the total norm (a
        device scalar; no host sync); the gradients are scaled where step() reads them, not in place."""
        pl = self._plan()
        if pl is None:
            return torch.zeros((), device=self._dev or 'cpu')
        self._prepare(pl, max_norm)
        return self._norm[0]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        pl = self._plan()
        if pl is None:
            return loss
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()
        elif self._hp_sent is None:
            raise RuntimeError('dsrg_amd.optim.AdamW: call sync_hyper() (or run one eager step) before capturing a step')
        if self._pending is not pl:
            self._prepare(pl, None)
        self._pending = None
        b1, b2, eps, cb = pl.consts
        _lib.check(_lib.load().dsrg_adamw_step(pl.table, pl.blocks, pl.n_blocks, _p(self._hp), _p(self._norm), _p(self._steps),
                                               b1, b2, eps, int(cb), _stream()), 'dsrg_adamw_step')
        self._gstash.clear()            # the handed-over gradients belong to the backward that just ended
        return loss

    def zero_grad(self, set_to_none=True):
        self._gstash.clear()
        return super().zero_grad(set_to_none)
