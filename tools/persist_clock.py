#!/usr/bin/env python3
"""Shader clock and per-workgroup span of the persistent 20-layer launch in the real sampler loop (BSG_PERSIST=1)."""
import ctypes
import os
import sys

import numpy as np
import torch

os.environ.setdefault('BSG_PERSIST', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bisinger_amd import _lib  # noqa: E402

torch.set_grad_enabled(False)
m = bench.build_model(torch.device('cuda', 0))
B, T = int(os.environ.get('PB', 16)), 1000
cond = torch.randn(B, 256, T, device='cuda')
x = m.philox_normal((B, 1, 80, T), 'cuda', 1, 0, 0)
m.sample(cond, x, seed=1, n_steps=int(os.environ.get('PN', 60)))
torch.cuda.synchronize()
n = min(512, B * 32)
buf = (ctypes.c_uint64 * (n * 4))()
_lib.check(_lib.load().bsg_diffnet_persist_clocks(m.denoise_fn._h, buf, n), 'clocks')
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.float64)
clk = (a[:, 2] - a[:, 0]) / (a[:, 3] - a[:, 1]) * 100
span = (a[:, 3] - a[:, 1]) / 100
print(f'B={B}: shader clock MHz median {np.median(clk):.0f} p10 {np.percentile(clk, 10):.0f} p90 {np.percentile(clk, 90):.0f}')
print(f'workgroup span us: p10 {np.percentile(span, 10):.0f} p50 {np.percentile(span, 50):.0f} p90 {np.percentile(span, 90):.0f} max {span.max():.0f}')
print(f'kernel span us {(a[:, 3].max() - a[:, 1].min()) / 100:.0f}; start spread us {(a[:, 1].max() - a[:, 1].min()) / 100:.1f}; timeouts {m.denoise_fn.handoff_timeouts()}')
