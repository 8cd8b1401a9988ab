#!/usr/bin/env python3
"""Shader clock held inside the stack launch at small batch sizes (the part forms stamp s_memtime / s_memrealtime like the one-workgroup
launch): python tools/clock_small.py [B ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
model = bench.build_model(dev)
net = model.denoise_fn
for B in [int(v) for v in (sys.argv[1:] or ['1', '4', '16'])]:
    wl = bench.Workload(model, dev, B, 0, 1)
    wl.step(1)
    net.profile(True)
    wl.step(2)
    torch.cuda.synchronize()
    mhz, span = net.clock_read()
    net.profile(False)
    print(f'B={B}: path {net.last_path()}  shader clock {mhz:.0f} MHz over a launch of {span:.0f} us')
