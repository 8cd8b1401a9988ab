"""Utterance sharding across the GPUs of one node + the path's single exchange step.

One process per GPU (torch.distributed, backend 'nccl' = RCCL over xGMI on ROCm; 'gloo' in the CPU
tests).  The reference has no inference-time parallelism; what shards is the batch:
rank r generates rows [r*B/g, (r+1)*B/g) with a full weight replica, and the only collective is one
all-gather of the finished mels ([B/g, T, 80] fp32 per rank — 5 MB at 16x1000x80, latency-bound).
Two things keep a sharded run equal to the unsharded one: the token-level front (ESM attends over the
batch axis) is evaluated on the whole batch by every rank, and the sampler noise is indexed by the
global row.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def init_distributed(backend=None):
    """Initialise the default process group from the torchrun environment.  Returns (rank, local_rank, world)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        if backend == 'nccl':
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_rows(B_total, rank, world):
    """Contiguous, balanced slice of the batch rows owned by ``rank`` (first B_total % world ranks get one more)."""
    base, rem = divmod(B_total, world)
    start = rank * base + min(rank, rem)
    return slice(start, start + base + (1 if rank < rem else 0))


def all_gather_rows(local, B_total, world, rank=None):
    """All-gather row shards [b_r, ...] -> [B_total, ...] in rank order (RCCL all-gather when shards are equal,
    padded all-gather otherwise)."""
    if world == 1:
        return local
    local = local.contiguous()
    counts = [shard_rows(B_total, r, world) for r in range(world)]
    sizes = [s.stop - s.start for s in counts]
    if len(set(sizes)) == 1:
        out = local.new_empty((B_total,) + tuple(local.shape[1:]))
        dist.all_gather_into_tensor(out, local)
        return out
    mx = max(sizes)
    pad = local.new_zeros((mx,) + tuple(local.shape[1:]))
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0)


def sharded_mel_gen(generate, B_total, rank, world):
    """``generate(rows: slice) -> mel [len(rows), T, M]`` for this rank's rows; returns the full [B_total, T, M]."""
    rows = shard_rows(B_total, rank, world)
    local = generate(rows)
    return all_gather_rows(local, B_total, world, rank)
