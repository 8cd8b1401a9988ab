#!/usr/bin/env python3
"""Per-WAVE phase timeline of residual_stack_q_kernel (BSG_STAMP_MODE=5: every wave stamps realtime + shader cycles; slots 1 / 2 mark the image
phase's inner boundaries).  Which wave is last at each barrier, and what it was doing: the critical path of a layer.
    BSG_STAMP_MODE=5 python tools/wave_stamps.py [B T]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bisinger_amd import _lib  # noqa: E402

torch.set_grad_enabled(False)
assert os.environ.get('BSG_STAMP_MODE') in ('5', '6')   # 6: + every weight reload from one L1-resident k-step (timing only)
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 1000)
dev = torch.device('cuda', 0)
m = bench.build_model(dev)
net = m.denoise_fn
cond = torch.randn(B, 256, T, device=dev)
x = torch.randn(B, 1, 80, T, device=dev)
t = torch.full((B,), 50, device=dev, dtype=torch.long)
for _ in range(3):
    net(x, t, cond)
torch.cuda.synchronize()
assert net.last_path() in ('stack_h2q', 'stack_h2'), net.last_path()
print('path', net.last_path())
tiles, L = B * ((T + 63) // 64), 20
D3 = os.environ.get('BSG_H2Q_DIAG') == '3'     # 32 slots per wave: + a stamp per pass of the GEMM loops
NS = 32 if D3 else 8
st = torch.zeros(tiles * L * 8 * NS, dtype=torch.int64, device=dev)
for rep in range(2):
    st.zero_()
    for _ in range(int(os.environ.get('STAMP_WARM', '500'))):
        net(x, t, cond)
    _lib.check(_lib.load().bsg_diffnet_debug_stack_stamps(net._h, 50, B, T, _lib.ptr(st), _lib.stream_ptr()), 'stamps')
    torch.cuda.synchronize()
raw = st.cpu().numpy().reshape(tiles, L, 8, NS)          # [tile][layer][wave][slot]
us = (raw & 0xffffffff).astype(np.float64) / 100.0
cy = ((raw >> 32) & 0xffffffff).astype(np.float64)
# slots: 0 layer start (behind the s1 scaling), 3 GEMM1 done, 4 barrier B passed, 5 GEMM2 + update done, 1 conditioner loads issued, 2 image rows written,
# 6 barrier C1 passed, 7 flag stored
inner = slice(2, L - 1)
def rel(a, slot, base):   # mean over tiles / layers of (stamp of each wave - base), per wave
    return (a[:, inner, :, slot] - base[:, inner, None]).mean((0, 1))
for name, a, unit in (('us', us, 'us'), ('cycles', cy, 'cyc')):
    t0 = a[:, :, :, 0].max(2)            # the layer starts for the last wave
    rB = a[:, :, :, 4].min(2)            # release of barrier B (first wave through)
    rC1 = a[:, :, :, 6].min(2)
    nxt = np.roll(a[:, :, :, 0].max(2), -1, axis=1)
    print(f'---- {name} (mean over tiles and inner layers; waves 0..7; wave w and w + 4 share a SIMD)')
    print(f'  layer period: {(nxt - t0)[:, inner].mean():.1f} {unit}')
    fmt = (lambda v: ' '.join(f'{x:7.2f}' for x in v)) if unit == 'us' else (lambda v: ' '.join(f'{x:7.0f}' for x in v))
    print('  layer start per wave, relative to the last wave\'s      ', fmt(rel(a, 0, t0)))
    print('  GEMM1 done (slot 3) - layer start of the last wave     ', fmt(rel(a, 3, t0)))
    print(f'  barrier B released - layer start                        {(rB - t0)[:, inner].mean():.2f}   (gate of the last wave: {(rB - a[:, :, :, 3].max(2))[:, inner].mean():.2f})')
    print('  GEMM2 + update done (slot 5) - barrier B release        ', fmt(rel(a, 5, rB)))
    print('  conditioner loads issued (slot 1) - barrier B release   ', fmt(rel(a, 1, rB)))
    print('  image rows written (slot 2) - barrier B release         ', fmt(rel(a, 2, rB)))
    print(f'  barrier C1 released - barrier B release                 {(rC1 - rB)[:, inner].mean():.2f}')
    print('  flag stored (slot 7, wave 0) - barrier C1 release       ', f'{(a[:, :, 0, 7] - rC1)[:, inner].mean():.2f}')
    print(f'  next layer start (last wave) - barrier C1 release       {(nxt - rC1)[:, inner].mean():.2f}')

if D3:
    rB = cy[:, :, :, 4].min(2)
    t0 = cy[:, :, :, 0].max(2)
    f = lambda v: ' '.join(f'{x:7.0f}' for x in v)
    print('---- GEMM loops, pass by pass (cycles; a pass = two k-steps = 96 MFMAs = 1536 matrix cycles per wave)')
    for it in range(12):
        print(f'  GEMM1 pass {it:2d} starts - layer start  ', f(rel(cy, 16 + it, t0)))
    print('  GEMM1 done                         ', f(rel(cy, 3, t0)))
    for it in range(4):
        print(f'  GEMM2 pass {it} starts - barrier B     ', f(rel(cy, 8 + it, rB)))
    print('  GEMM2 loop done                    ', f(rel(cy, 12, rB)))
    print('  next GEMM1 fragments requested     ', f(rel(cy, 13, rB)))
    print('  x / skip updated (slot 5)          ', f(rel(cy, 5, rB)))
if os.environ.get('PER_ROW'):
    tpr = (T + 63) // 64
    rC1 = cy[:, :, :, 6].min(2)
    w = (cy[:, :, 0, 7] - rC1)[:, inner].reshape(B, tpr, -1).mean((1, 2))
    print('  per-row flag stored - barrier C1 release (cycles):', ' '.join(f'{v:.0f}' for v in w))
    st10 = us[:, 10, :, 0].max(1).reshape(B, tpr)
    print('  per-row start of layer 10 (us, rel.):', ' '.join(f'{v:.1f}' for v in (st10.mean(1) - st10.min())))
    print('  even / odd tiles of row 0, start of layer 10 (us):', ' '.join(f'{v:.1f}' for v in (st10[0] - st10.min())))
