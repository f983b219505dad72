"""gpu: bench.py's launch contract on the one-GPU box.  `python bench.py --gpus 2` (no launcher) must start two ranks
itself and print ONE JSON line with n_gpus == 2; the ranks rendezvous over gloo and share the visible device
(DSRG_BENCH_BACKEND=gloo: the rehearsal mode; on a multi-GPU node the same command runs one rank per GPU over RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_from_the_gpus_flag():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    env['DSRG_BENCH_BACKEND'] = 'gloo'
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '2',
                        '--no-cpu-baseline', '--no-roofline'], env=env, capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 6 and out['scaling'] == 'weak'
    assert out['value'] > 0 and out['config']['episodes_per_gpu'] == 64


def test_pretraining_step_two_ranks_ddp_rehearsal():
    """`bench.py --config c4 --gpus 2` on the one-GPU box: two ranks of the HIP pre-training model under
    DistributedDataParallel (find_unused_parameters, task-id broadcast, gradient all-reduce) run whole optimisation steps
    and agree on finite losses.  Rehearsal backend: gloo, gradient buckets staged through host memory
    (dsrg_amd.distributed.cpu_staged_allreduce_hook); on a multi-GPU node the same command runs over RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    env['DSRG_BENCH_BACKEND'] = 'gloo'
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--config', 'c4', '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--no-cpu-baseline', '--no-roofline'], env=env, capture_output=True, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['name'] == 'c4' and out['value'] > 0
    assert set(out['final_losses']) == {'mlm', 'mrc', 'sap'}
    assert all(v == v and abs(v) < 1e4 for v in out['final_losses'].values())
