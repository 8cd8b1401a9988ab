#!/usr/bin/env python3
"""HiFi-GAN generator alone: ms per forward at (B, T) pairs; TFLOP/s by the algorithmic 38.51 MFLOP per mel frame."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
voc, cfg = bench.build_vocoder(dev)
out = {}
for B, T in ((1, 1000), (8, 1000), (16, 1000)):
    mel = torch.randn(B, 80, T, device=dev)
    voc(mel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        w = voc(mel)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    out[f'B{B}'] = {'ms': round(dt * 1e3, 3), 'tflops': round(38.51e6 * B * T / dt / 1e12, 1), 'finite': bool(torch.isfinite(w).all())}
print(json.dumps(out))
