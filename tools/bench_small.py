#!/usr/bin/env python3
"""Latency sweep: ms per 100-step mel-generation pass at T = 1000 for a list of batch sizes (the channel-split launch paths at
B = 1..16; BSG_DTYPE=bf16 sweeps the bf16-operand configuration, e.g. the stack launch against per-layer launches)."""
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
model = bench.build_model(dev)
model.denoise_fn.set_compute('bf16' if os.environ.get('BSG_DTYPE') == 'bf16' else 'fp32')
out = {}
for B in [int(v) for v in (sys.argv[1:] or ['1', '2', '3', '4', '6', '8', '12', '16'])]:
    wl = bench.Workload(model, dev, B, 0, 1)
    wl.step(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3):
        wl.step(2 + i)
    torch.cuda.synchronize()
    out[B] = {'ms_per_pass': round((time.perf_counter() - t0) / 3 * 1e3, 2), 'path': model.denoise_fn.last_path()}
print(json.dumps(out))
