#!/usr/bin/env python3
"""Micro-benchmark: the fused residual-layer kernel and a whole DiffNet call (GPU box)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bisinger_amd import synth  # noqa: E402
from tests.util import load_formula_weights, use_config  # noqa: E402

torch.set_grad_enabled(False)
use_config()
from bisinger_amd.diffnet import DiffNet  # noqa: E402

net = load_formula_weights(DiffNet(80), 0, synth.DIFFNET_GAIN, prefix='denoise_fn.').cuda()
DTYPE = sys.argv[1] if len(sys.argv) > 1 else 'fp32'   # fp32 | bf16
net.set_compute(DTYPE)
FLOP_LAYER = 1048576.0
FLOP_NET = 21184512.0
SHAPES = [(16, 1000), (64, 1000), (8, 1000), (1, 500)]
if len(sys.argv) > 2:   # e.g. 1x1000,2x1000,4x1000
    SHAPES = [tuple(int(v) for v in p.split('x')) for p in sys.argv[2].split(',')]
for B, T in SHAPES:
    cond = torch.randn(B, 256, T, device='cuda')
    x = torch.randn(B, 256, T, device='cuda')
    skip = torch.zeros(B, 256, T, device='cuda')
    t = torch.full((B,), 50, dtype=torch.long, device='cuda')
    net.prepare(cond)
    for nb in [1]:
        for _ in range(3):
            net.residual_layer(3, x, t, skip)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            net.residual_layer(3, x, t, skip)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(f'{DTYPE} layer B={B} T={T}: {ms * 1e3:.1f} us  {FLOP_LAYER * B * T / ms / 1e9:.1f} TFLOP/s  {6144.0 * B * T / ms / 1e6:.0f} GB/s (algorithmic)', flush=True)
    spec = torch.randn(B, 1, 80, T, device='cuda')
    for _ in range(2):
        net(spec, t, cond)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        net(spec, t, cond)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f'diffnet B={B} T={T}: {ms:.3f} ms  {FLOP_NET * B * T / ms / 1e9:.1f} TFLOP/s  -> {B * T / (ms * 100 / 1e3):.0f} frames/s @100 steps', flush=True)
