#!/usr/bin/env python3
"""What the part forms' exchange WAITS cost: the same sampler launches with every hand-off wait skipped (bsg_diffnet_debug_inject_giveup with a
negative count: the next n part launches do not poll and count nothing — wrong results, the same instructions otherwise), timed beside the normal launches with HIP
events around bsg_ddpm_sample (direct ABI: the Python guard would repeat a call that counted give-ups).  PB = batch (8: the pair form of the
configs[3] rank; 1: the quads of a single utterance), T = 1000, 100 steps."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = sys.argv[:1]
import bench  # noqa: E402
from bisinger_amd import _lib  # noqa: E402
from ctypes import byref  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
model = bench.build_model(dev)
net = model.denoise_fn
lib = _lib.load()
B, T = int(os.environ.get('PB', 8)), 1000
cond = torch.randn(B, 256, T, device=dev)
x0 = torch.randn(B, 1, 80, T, device=dev)
model.sample(cond, x0.clone(), seed=1)          # binds, allocates, warms up
path = net.last_path()
s, _keep = model._schedule()
h = net._h


def run(n_pass, inject):
    ts = []
    for i in range(n_pass):
        x = x0.clone()
        if inject:
            net.debug_inject_giveup(-100)       # negative: the next 100 part launches skip their waits silently
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(lib.bsg_ddpm_sample(h, byref(s), _lib.ptr(x), None, 7, 99, 100, B, T, 0, B, _lib.stream_ptr()), 'bsg_ddpm_sample')
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
        net.take_health()
    ts.sort()
    return ts[len(ts) // 2]


for rep in range(2):
    normal = run(5, False)
    nowait = run(5, True)
    print(f'B={B} {path}: 100 sampler steps {normal:.2f} ms; with every hand-off wait skipped {nowait:.2f} ms ({(normal - nowait) / normal * 100:.1f} % of the loop '
          f'= {(normal - nowait) * 10 / 20:.2f} us per layer)')
