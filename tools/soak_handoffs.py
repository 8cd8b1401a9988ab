#!/usr/bin/env python3
"""Soak test of the launches that hand data between workgroups (channel-split launches at small batch; the stack
launches from B = 5): repeated 100-step sampler runs must be bit-identical and report zero hand-off time-outs."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

torch.set_grad_enabled(False)
m = bench.build_model(torch.device('cuda', 0))
n_rep = int(os.environ.get('SOAK_REPS', 6))
for B, T in [(1, 1000), (2, 777), (4, 1000), (3, 250), (6, 1000), (8, 1000), (16, 1000)]:
    cond = torch.randn(B, 256, T, device='cuda')
    x0 = torch.randn(B, 1, 80, T, device='cuda')
    ref = m.sample(cond, x0.clone(), seed=3).clone()
    ok = True
    for _ in range(n_rep):
        ok &= bool(torch.equal(m.sample(cond, x0.clone(), seed=3), ref))
    torch.cuda.synchronize()
    print(f'B={B} T={T}: {n_rep} repeats identical={ok} finite={bool(torch.isfinite(ref).all())} '
          f'handoff_timeouts={m.denoise_fn.handoff_timeouts()}', flush=True)
    assert ok and m.denoise_fn.handoff_timeouts() == 0
print('soak ok')
