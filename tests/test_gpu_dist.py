"""GPU: RCCL comes up and runs the path's one collective on the driver box (world_size 1: the box has one GPU; the
multi-rank logic is covered over gloo in tests/test_dist_cpu.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, json, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from bisinger_amd import dist as bdist
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
torch.cuda.set_device(0)
dist.init_process_group(backend='nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
x = torch.arange(2 * 5 * 80, dtype=torch.float32, device='cuda').reshape(2, 5, 80)
out = torch.empty_like(x)
dist.all_gather_into_tensor(out, x)            # the collective bench.py / sharded_mel_gen issue (RCCL all-gather)
full = bdist.sharded_mel_gen(lambda rows: x[rows], 2, 0, 1)
t = torch.tensor([3.0], device='cuda', dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)       # bench.py's max-over-ranks timing
dist.barrier()
torch.cuda.synchronize()
print(json.dumps({'ok': bool(torch.equal(out, x) and torch.equal(full, x) and float(t) == 3.0),
                  'backend': dist.get_backend(), 'world': dist.get_world_size()}))
dist.destroy_process_group()
'''


def test_rccl_world1_all_gather():
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29611', HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, '-c', CHILD % ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    import json
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])
    assert rec == {'ok': True, 'backend': 'nccl', 'world': 1}


def test_bench_self_launch_two_ranks_on_one_gpu_is_refused_cleanly():
    """python bench.py --gpus 2 from a bare shell starts its own workers; on a 1-GPU box it must say so, not hang or crash"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.device_count() >= 2:
        assert p.returncode == 0, p.stderr[-2000:]
    else:
        assert p.returncode != 0 and 'GPUs' in (p.stderr + p.stdout)
