#!/usr/bin/env python3
"""Phase timeline of the fused residual-layer kernel from the diagnostic (stamped) build."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bisinger_amd import _lib, synth  # noqa: E402
from tests.util import load_formula_weights, use_config  # noqa: E402

torch.set_grad_enabled(False)
use_config()
from bisinger_amd.diffnet import DiffNet  # noqa: E402

B, T = int(os.environ.get('PB', 16)), int(os.environ.get('PT', 1000))
net = load_formula_weights(DiffNet(80), 0, synth.DIFFNET_GAIN, prefix='denoise_fn.').cuda()
DT = os.environ.get('PD', 'fp32')
net.set_compute(DT)
cond = torch.randn(B, 256, T, device='cuda')
x = torch.randn(B, 256, T, device='cuda')
out = torch.empty_like(x)
skip = torch.zeros(B, 256, T, device='cuda')
t = torch.full((B,), 50, dtype=torch.long, device='cuda')
net.prepare(cond)
NTILE = 64 if DT == 'bf16' else 32
nwg = B * ((T + NTILE - 1) // NTILE)
st = torch.zeros(nwg, 8, 10, dtype=torch.int64, device='cuda')
lib = _lib.load()
for it in range(3):
    for _ in range(200):
        net.residual_layer(3, x, t, skip)
    _lib.check(lib.bsg_diffnet_debug_stamps(net._h, 3, _lib.ptr(x), _lib.ptr(t), _lib.ptr(out), _lib.ptr(skip), B, T, _lib.ptr(st),
                                            _lib.stream_ptr()), 'stamps')
    torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.float64)
t0 = s[:, :, 0].min()
names = ['start', 'staged+barrier', 'acc-init issued', 'GEMM1 done', 'gate done', 'z in LDS (2 barriers)', 'GEMM2 residual', 'GEMM2 skip']
print(f'B={B} T={T} workgroups={nwg}; s_memtime ticks (100 MHz constant clock if memrealtime, else shader cycles)')
print('kernel span (first start -> last end):', s[:, :, 7].max() - t0)
print('wg start spread:', s[:, :, 0].min(axis=1).max() - t0)
clk = (s[:, :, 7] - s[:, :, 0]) / np.maximum(s[:, :, 9] - s[:, :, 8], 1) * 100.0
print(f'shader clock held (MHz): median {np.median(clk):.0f}  p10 {np.percentile(clk, 10):.0f}  p90 {np.percentile(clk, 90):.0f}')
rt = s[:, :, 8:10]
print(f'kernel span by the 100 MHz clock: {(rt[:, :, 1].max() - rt[:, :, 0].min()) / 100.0:.1f} us')
wg_start = rt[:, :, 0].min(axis=1); wg_end = rt[:, :, 1].max(axis=1); k0 = wg_start.min()
print(f'workgroup start (us after first): p50 {np.percentile(wg_start - k0, 50) / 100:.2f}  p90 {np.percentile(wg_start - k0, 90) / 100:.2f}  max {(wg_start - k0).max() / 100:.2f}')
print(f'workgroup end   (us after first start): p10 {np.percentile(wg_end - k0, 10) / 100:.1f}  p50 {np.percentile(wg_end - k0, 50) / 100:.1f}  p90 {np.percentile(wg_end - k0, 90) / 100:.1f}  max {(wg_end - k0).max() / 100:.1f}')
print(f'workgroup duration us: p10 {np.percentile(wg_end - wg_start, 10) / 100:.1f} p50 {np.percentile(wg_end - wg_start, 50) / 100:.1f} p90 {np.percentile(wg_end - wg_start, 90) / 100:.1f}')
order = np.argsort(wg_end)
print('last 8 workgroups to finish (blockIdx, start us, end us):', [(int(i), round(float(wg_start[i] - k0) / 100, 1), round(float(wg_end[i] - k0) / 100, 1)) for i in order[-8:]])
d = np.diff(s[:, :, :8], axis=2)
for i in range(7):
    print(f'  phase {i}->{i + 1} {names[i + 1]:28s} mean {d[:, :, i].mean():10.0f}  p10 {np.percentile(d[:, :, i], 10):10.0f}  p90 {np.percentile(d[:, :, i], 90):10.0f}')
print('  per-wave total mean', (s[:, :, 7] - s[:, :, 0]).mean())

# intra-workgroup skew: how long the first wave to finish GEMM1 waits for the last one of its workgroup
g1 = s[:, :, 3]
skew = g1.max(axis=1) - g1.min(axis=1)
early = wg_end < np.median(wg_end)
for name, sel in (('older half (finishes first)', early), ('younger half', ~early)):
    print(f'GEMM1-end skew inside a workgroup, {name}: mean {skew[sel].mean():.0f} p50 {np.percentile(skew[sel], 50):.0f} p90 {np.percentile(skew[sel], 90):.0f} cycles;'
          f' GEMM1 duration mean {(s[sel][:, :, 3] - s[sel][:, :, 2]).mean():.0f}; barrier phase mean {(s[sel][:, :, 5] - s[sel][:, :, 4]).mean():.0f}')
# per-wave-slot pattern: which waves finish GEMM1 last
rank = np.argsort(np.argsort(g1, axis=1), axis=1).mean(axis=0)
print('mean finishing rank of wave 0..7 (0 = first):', np.round(rank, 2))
