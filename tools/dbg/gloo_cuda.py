import os, sys, time, torch, torch.distributed as dist, torch.multiprocessing as mp
def run(rank):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29611', RANK=str(rank), WORLD_SIZE='2')
    import faulthandler; faulthandler.dump_traceback_later(100, exit=True)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo')
    for n in (1 << 10, 1 << 20, 25 << 20):
        t = torch.ones(n, device='cuda')
        torch.cuda.synchronize(); t0 = time.time()
        dist.all_reduce(t); torch.cuda.synchronize()
        print(rank, 'allreduce', n * 4 >> 20, 'MiB', '%.3f s' % (time.time() - t0), float(t[0]), flush=True)
    m = torch.nn.Sequential(*[torch.nn.Linear(4096, 4096) for _ in range(int(os.environ.get('NL', '1')))]).cuda()
    print(rank, 'params MB', sum(p.numel() for p in m.parameters()) * 4 >> 20, flush=True)
    ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], find_unused_parameters=True, gradient_as_bucket_view=True, bucket_cap_mb=100)
    ddp(torch.randn(8, 4096, device='cuda')).sum().backward(); torch.cuda.synchronize()
    print(rank, 'ddp ok', flush=True)
    dist.destroy_process_group()
if __name__ == '__main__':
    ps = [mp.get_context('spawn').Process(target=run, args=(r,)) for r in range(2)]
    [p.start() for p in ps]; [p.join() for p in ps]
