"""Data-parallel plumbing for the DSRG hot path (one process per GPU, torch.distributed; backend 'nccl' is RCCL
on ROCm and runs over xGMI, 'gloo' for CPU tests).

The path shards ONLY across the episode batch:
  * inference / evaluation: replicas, rank r takes a contiguous slice of the episodes (the reference splits its
    validation data the same way, finetune_src/r2r/env.py:110-118); no collective on the data path, results are
    gathered once at the end (finetune_src/utils/distributed.py:90-130 `all_gather` of pickled objects);
  * training: DDP gradient all-reduce, `find_unused_parameters=True` because heads / text-side sub-layers are
    unused per task or mode (pretrain_src/utils/misc.py:52-65, finetune_src/r2r/agent_base.py:118-120), plus a
    1-int64 task-id broadcast per pre-training step (pretrain_src/data/loader.py:47-51).
"""
import os
import torch
import torch.distributed as dist


def init_distributed(backend=None, device=None):
    """env:// rendezvous (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from torchrun).  Returns (rank, world)."""
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
        kw = {'device_id': device} if (backend == 'nccl' and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_slice(n, rank, world):
    """contiguous [lo, hi) of n episodes for this rank; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(batch, rank, world, batch_size=None):
    """Slice every per-episode entry (tensors / arrays / lists whose leading dim is the batch) of a mode batch."""
    if batch_size is None:
        batch_size = next(v.shape[0] for v in batch.values() if hasattr(v, 'shape') and getattr(v, 'ndim', 0) > 0)
    lo, hi = shard_slice(batch_size, rank, world)
    out = {}
    for k, v in batch.items():
        if hasattr(v, 'shape') and getattr(v, 'ndim', 0) > 0 and v.shape[0] == batch_size:
            out[k] = v[lo:hi]
        elif isinstance(v, (list, tuple)) and len(v) == batch_size:
            out[k] = v[lo:hi]
        else:
            out[k] = v
    return out


def all_gather_objects(obj):
    """Gather arbitrary picklable per-rank results (validation only; never on the hot path)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def broadcast_task_id(task_id, device='cpu'):
    """rank 0 picks the task, everyone follows (pretrain_src/data/loader.py:47-51)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return int(task_id)
    if dist.get_backend() == 'gloo':
        device = 'cpu'
    t = torch.tensor([int(task_id)], dtype=torch.int64, device=device)
    dist.broadcast(t, 0)
    return int(t.item())


def bf16_compress_hook(process_group, bucket):
    """DDP comm hook: all-reduce gradients in bf16 (halves xGMI bytes: 981 MB -> 490 MB per step for the 245 M
    parameter pre-training model), decompress into the fp32 bucket.  bf16 keeps fp32's exponent range, so no
    loss scaling is needed."""
    group = process_group if process_group is not None else dist.group.WORLD
    world = dist.get_world_size(group)
    buf = bucket.buffer()
    comp = buf.to(torch.bfloat16).div_(world)
    fut = dist.all_reduce(comp, group=group, async_op=True).get_future()

    def decompress(f):
        buf.copy_(f.value()[0])
        return buf
    return fut.then(decompress)


def cpu_staged_allreduce_hook(process_group, bucket):
    """DDP comm hook for the REHEARSAL backend only (gloo with the model on a GPU, e.g. two ranks sharing the one GPU of a
    test box): gloo's own path for large CUDA tensors stalls on this pool (a plain 768 MB nn.Sequential under DDP stalls in
    its first broadcast, tools/dbg/gloo_cuda.py), so the bucket is staged through host memory and reduced there.  Slow and
    synchronous by design; RCCL runs never take it."""
    group = process_group if process_group is not None else dist.group.WORLD
    buf = bucket.buffer()
    cpu = buf.detach().to('cpu', torch.float32)
    dist.all_reduce(cpu, group=group)
    buf.copy_(cpu.div_(dist.get_world_size(group)).to(buf.dtype))
    fut = torch.futures.Future()
    fut.set_result(buf)
    return fut


def wrap_model(model, device=None, bf16_grads=False, bucket_cap_mb=100, static_graph=False, force=False):
    """DDP with the reference's semantics.  Larger buckets than torch's 25 MB default: xGMI links are point to point
    (~153 GB/s each), so fewer, larger all-reduces amortise the per-collective latency."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):      # force: wrap even a single rank (tests)
        return model
    from torch.nn.parallel import DistributedDataParallel as DDP
    from . import functional as _Fn
    _Fn.MATERIALIZE_UNUSED_GRADS = True        # see functional._PackFn
    ids = [device.index] if (device is not None and device.type == 'cuda') else None
    rehearsal = dist.get_backend() == 'gloo' and device is not None and device.type == 'cuda'
    # rehearsal: no initial broadcast / buffer sync through gloo's CUDA path either (see cpu_staged_allreduce_hook); the
    # caller loads the same weights on every rank
    extra = {'init_sync': False, 'broadcast_buffers': False} if rehearsal else {}
    ddp = DDP(model, device_ids=ids, find_unused_parameters=not static_graph, static_graph=static_graph,
              bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True, **extra)
    if rehearsal:
        ddp.register_comm_hook(None, cpu_staged_allreduce_hook)
    elif bf16_grads:
        ddp.register_comm_hook(None, bf16_compress_hook)
    return ddp


def pin_step_threads(local_rank=0, local_world=1):
    """Put the calling thread - and the autograd engine's device thread, if it is created afterwards (it inherits the mask) -
    on ONE core of this process's allowed set, a different one per local rank.  A training step of this path is launch-bound
    Python: the main thread and the engine thread never run at the same time, and when the scheduler places them on two cores
    every forward -> backward -> optimizer hand-over pays a cross-core wake-up on cold caches (measured on the pre-training
    step: 18.4 ms per step on two cores, 16.1 on one, same binary, same box).  Threads that already exist (RCCL / gloo
    progress threads, created by init_process_group) keep their mask.  Returns the previous mask (restore with
    os.sched_setaffinity(0, mask)), or None where the platform has no affinity call.
    Inheritance cuts both ways: EVERY thread created while the pin is in force starts on that one core - intra-op pools
    (torch.set_num_threads takes effect at the first parallel op), DataLoader workers, a CPU baseline.  Create such pools
    before calling this (bench.py warms torch's pool first), or restore the mask before creating them."""
    if not hasattr(os, 'sched_getaffinity'):
        return None
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) <= 1:
        return set(allowed)
    stride = max(1, len(allowed) // max(1, local_world))
    core = allowed[(local_rank * stride + stride // 2) % len(allowed)]
    os.sched_setaffinity(0, {core})
    return set(allowed)


def reduce_max(x, device='cpu'):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
