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


RANK_CHILD = r'''
import os, sys, json, warnings, torch
sys.path.insert(0, %r)
sys.argv = sys.argv[:1]
import numpy as np
import torch.distributed as dist
import bench
from bisinger_amd import dist as bdist, synth
torch.set_grad_enabled(False)
rank, _, world = bdist.env_world()
dist.init_process_group(backend='gloo', rank=rank, world_size=world)        # both ranks compute on THE one GPU of the box: RCCL cannot
torch.cuda.set_device(0)                                                    # place two ranks on one device, gloo carries the exchange
dev = torch.device('cuda', 0)
model = bench.build_model(dev)
B, Tt, T = 6, 12, 120
inp = {k: torch.from_numpy(v).to(dev) for k, v in synth.synth_inputs(B, Tt, T, seed=4, ragged=True).items()}
kw = {k: inp[k] for k in ('pitch_midi', 'midi_dur', 'is_slur', 'lang', 'speechsing')}

def generate(rows):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')        # two processes share the chip: a hand-off may give up and heal inside the call
        out = model(inp['txt_tokens'], mel2ph=inp['mel2ph'], spk_embed=inp['spk_embed'], infer=True, seed=5, rows=rows, **kw)
    return out['mel_out'].cpu()

full = bdist.sharded_mel_gen(generate, B, rank, world)
dist.barrier()
if rank == 0:
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        ref = model(inp['txt_tokens'], mel2ph=inp['mel2ph'], spk_embed=inp['spk_embed'], infer=True, seed=5, **kw)['mel_out'].cpu()
    print(json.dumps({'shape': list(full.shape), 'max_abs': float((full - ref).abs().max()), 'finite': bool(torch.isfinite(full).all()),
                      'pending_timeouts': model.denoise_fn.handoff_timeouts()}), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_run_the_hip_path_and_gather():
    """two rank PROCESSES, both generating their utterance shard on the HIP path (sharing the box's one GPU — the
    situation the hand-off guards exist for), exchanged with bisinger_amd.dist's all-gather over gloo: the gathered batch equals the
    unsharded run (Philox noise by global row, token-level front on the whole batch) to the rounding of the launch forms."""
    import json
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, '-c', RANK_CHILD % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    rec = json.loads([l for l in outs[0][0].splitlines() if l.startswith('{')][-1])
    assert rec['shape'] == [6, 120, 80] and rec['finite'] and rec['pending_timeouts'] == 0
    assert rec['max_abs'] <= 1e-3, rec


def test_bench_n_gt_1_code_path_rehearsed_on_one_gpu():
    """The driver runs `bench.py --gpus N` (N = 2, 4, 8) on an 8-GPU node that has never been available to this build: the N > 1 branch — rows of
    configs[3] per rank with the rank-local token front, barrier + max-over-ranks timing, the gathered mels, the weak-scaling figure, the
    record with the part kernel named in `roofline` — is rehearsed here with TWO rank processes on the one GPU of the box
    (--rehearse-one-gpu: collectives over gloo through host memory).  Checks the record's shape, not its numbers."""
    import json
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1', '--rehearse-one-gpu'],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stderr or '')[-3000:]
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])
    assert rec['n_gpus'] == 2 and rec['scaling'] == 'strong' and rec['config']['global_batch'] == 64 and rec['config']['utterances_per_gpu'] == 32
    assert rec['metric'] == 'mel_frames_per_sec' and rec['value'] > 0 and rec['steps'] == 1 and rec['handoff_timeouts'] == 0
    assert rec['roofline']['frac'] > 0 and 'kernel' in rec['roofline']
    assert rec['weak_scaling']['global_batch'] == 32 and rec['weak_scaling']['finite']
    assert rec['collective_backend'] == 'gloo' and 'rehearsal' in rec
