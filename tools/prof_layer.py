#!/usr/bin/env python3
"""Profiling driver: N launches of the fused residual-layer kernel at the bench shape (B=16, T=1000).
Run under rocprofv3 (--kernel-trace --stats, or --pmc <counters>) on the GPU box."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bisinger_amd import synth  # noqa: E402
from tests.util import load_formula_weights, use_config  # noqa: E402

torch.set_grad_enabled(False)
use_config()
from bisinger_amd.diffnet import DiffNet  # noqa: E402

B = int(os.environ.get('PB', 16))
T = int(os.environ.get('PT', 1000))
N = int(os.environ.get('PN', 20))
net = load_formula_weights(DiffNet(80), 0, synth.DIFFNET_GAIN, prefix='denoise_fn.').cuda()
net.set_compute(os.environ.get('PD', 'fp32'))   # fp32 | bf16
cond = torch.randn(B, 256, T, device='cuda')
x = torch.randn(B, 256, T, device='cuda')
skip = torch.zeros(B, 256, T, device='cuda')
t = torch.full((B,), 50, dtype=torch.long, device='cuda')
net.prepare(cond)
for i in range(N):
    net.residual_layer(1 + i % 18, x, t, skip)
torch.cuda.synchronize()
print('done')
