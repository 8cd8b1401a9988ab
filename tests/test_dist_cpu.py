"""CPU, world_size 2 (gloo): utterance sharding + the all-gather reproduce the unsharded result.
The generator here is the oracle (the product path needs a GPU); what is under test is bisinger_amd.dist
and the sharding rule of SURVEY.md §8e (token-level front on the full batch, noise by global row)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from bisinger_amd import dist as bdist, synth


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, out_path):
    import json
    from collections import OrderedDict
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    torch.set_grad_enabled(False)
    r, lr, w = bdist.init_distributed('gloo')
    assert (r, w) == (rank, world)
    from oracle import diffusion as odf, fs2 as ofs2, melgen as omg
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = OrderedDict((k, tuple(s)) for k, s in json.load(open(os.path.join(root, 'tests/golden/state_dict_spec.json')))['GaussianDiffusion'])
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(spec, 0, synth.DIFFNET_GAIN).items()}
    sd.update(odf.make_schedule(8, 'linear', 0.06))
    g = np.load(os.path.join(root, 'tests/golden/schedules.npz'))
    sd['spec_min'], sd['spec_max'] = torch.from_numpy(g['spec_min']), torch.from_numpy(g['spec_max'])
    inp = {k: torch.from_numpy(v) for k, v in synth.synth_inputs(B, 8, 40, seed=4, ragged=True).items()}
    noise = torch.from_numpy(synth.synth_noise(8, B, 80, 40, seed=6))

    def generate(rows):
        fs2_out = ofs2.fs2_forward(sd, inp, rows=rows)
        sub = {k: v[rows] for k, v in inp.items()}
        return omg.mel_gen(sd, sub, noise[:, rows], timesteps=8, K_step=8, fs2_out=fs2_out)['mel_out']

    full = bdist.sharded_mel_gen(generate, B, rank, world)
    if rank == 0:
        ref = omg.mel_gen(sd, inp, noise, timesteps=8, K_step=8)['mel_out']
        naive = torch.cat([omg.mel_gen(sd, {k: v[bdist.shard_rows(B, q, world)] for k, v in inp.items()},
                                       noise[:, bdist.shard_rows(B, q, world)], timesteps=8, K_step=8)['mel_out'] for q in range(world)])
        np.savez(out_path, full=full.numpy(), ref=ref.numpy(), naive=naive.numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('B', [4, 5])
def test_sharded_equals_unsharded_world2(tmp_path, B):
    out = str(tmp_path / 'r.npz')
    mp.spawn(_worker, args=(2, _free_port(), B, out), nprocs=2, join=True)
    r = np.load(out)
    assert r['full'].shape == r['ref'].shape == (B, 40, 80)
    assert np.abs(r['full'] - r['ref']).max() <= 2e-5
    # sharding the token-level front as well would NOT reproduce the reference (ESM couples rows)
    assert np.abs(r['naive'] - r['ref']).max() > 1e-3


def test_rank_local_front_equals_the_whole_batch_front():
    """What bsg_fs2midi_encode_rows computes (restated in oracle.fs2.fs2_forward(local_front=True)): the ESM's K / V — projections of
    LN(lang_embed[lang]), common_layers.py:850-853 — from EVERY row of the batch, everything else of the token-level front on the rank's
    rows.  Against the reference's way (the front on the whole batch, sliced): the same rows to fp32 rounding — and a front that saw the
    rank's rows ALONE is off by far more than the parity bar."""
    from bisinger_amd import synth
    from oracle import fs2 as ofs2
    torch.manual_seed(0)
    import json
    spec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'state_dict_spec.json')))
    from collections import OrderedDict
    sdspec = OrderedDict((k, tuple(v)) for k, v in spec['GaussianDiffusion'] if k.startswith('fs2.'))
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(sdspec, 0).items()}
    B, Tt, T = 6, 20, 60
    inp = {k: torch.from_numpy(v) for k, v in synth.synth_inputs(B, Tt, T, seed=2, ragged=True).items()}
    inp['lang'] = torch.from_numpy(np.random.RandomState(1).randint(0, 2, (B, Tt)).astype(np.int64))
    for rows in (slice(0, 2), slice(2, 5), slice(5, 6)):
        ref = ofs2.fs2_forward(sd, inp, 'fs2.', rows=rows)
        loc = ofs2.fs2_forward(sd, inp, 'fs2.', rows=rows, local_front=True)
        alone = ofs2.fs2_forward(sd, {k: v[rows] for k, v in inp.items()}, 'fs2.')
        for k in ('decoder_inp', 'mel_out'):
            assert float((ref[k] - loc[k]).abs().max()) <= 2e-5, k
        assert torch.equal(ref['mel2ph'], loc['mel2ph'])
        assert float((ref['decoder_inp'] - alone['decoder_inp']).abs().max()) > 1e-3


def test_shard_rows_partition():
    for B in (1, 7, 16, 64):
        for w in (1, 2, 3, 8):
            sl = [bdist.shard_rows(B, r, w) for r in range(w)]
            assert sl[0].start == 0 and sl[-1].stop == B
            assert all(a.stop == b.start for a, b in zip(sl, sl[1:]))
            sizes = [s.stop - s.start for s in sl]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize('n', [2, 8])      # 8: the rank count of BASELINE configs[3] (the driver's SCALE run)
def test_bench_starts_its_own_workers_from_a_bare_shell(n):
    """`python bench.py --gpus N` without torchrun must start N fresh worker processes itself (the parent touches no GPU),
    hand them the torchrun environment and print rank 0's record as the LAST stdout line.  Here: the process-management
    self-test over gloo (no GPU in this container); without --selftest-procs it must refuse cleanly, not hang."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(n), '--selftest-procs'], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    last = [l for l in p.stdout.splitlines() if l.strip()][-1]
    assert json.loads(last) == {'selftest': True, 'n_gpus': n, 'ranks': list(range(n))}
    if not torch.cuda.is_available() and n == 2:
        p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2'], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode != 0 and 'GPUs' in p.stderr
