#!/usr/bin/env python3
"""Profiling driver: BASELINE configs[3] as ONE of its 8 ranks sees it (B_total = 64, token front on 64 rows, frames + sampler on
rows 8..15), PN passes (run under rocprofv3 --kernel-trace --stats on the GPU box; bench.py `secondary.cfg3_rank` is the timed form).
PB / PW: another (B_total, world) — PB=1 PW=1 profiles the single-utterance pass."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = sys.argv[:1]
import bench  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
model = bench.build_model(dev)
B, W = int(os.environ.get('PB', 64)), int(os.environ.get('PW', 8))
wl = bench.Workload(model, dev, B, 1 if W > 1 else 0, W, emulate=W > 1)
wl.step(1)
torch.cuda.synchronize()
n = int(os.environ.get('PN', 3))
t0 = time.perf_counter()
for i in range(n):
    mel = wl.step(2 + i)
torch.cuda.synchronize()
print(f'rank pass: {(time.perf_counter() - t0) / n * 1e3:.2f} ms  path={model.denoise_fn.last_path()}  mel={tuple(mel.shape)}')
