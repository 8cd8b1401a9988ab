#!/usr/bin/env python3
"""Phase timeline of the on-chip stack launches (split-fp16, F(4,3), bf16) from its in-kernel s_memrealtime stamps.
    python tools/stack_stamps.py [B T]      (default 16 1000)
Prints, averaged over tiles and layers, the duration of each phase of a layer, the layer period, and how the start of
layer 10 spreads over the tiles (are the two workgroups of a CU in phase?)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bisinger_amd import _lib  # noqa: E402

torch.set_grad_enabled(False)
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 1000)
dev = torch.device('cuda', 0)
m = bench.build_model(dev)
net = m.denoise_fn
DT = os.environ.get('PD', 'fp32')
net.set_compute(DT)
cond = torch.randn(B, 256, T, device=dev)
x = torch.randn(B, 1, 80, T, device=dev)
t = torch.full((B,), 50, device=dev, dtype=torch.long)
for _ in range(3):
    net(x, t, cond)          # binds, fills xa, warms up
torch.cuda.synchronize()
F43 = net.last_path() == 'stack_f43'
H2 = net.last_path() in ('stack_h2', 'stack_h2q')   # split-fp16 form: same stamp slots as the bf16 stack launch
PAIR = net.last_path() in ('stack_h2_quad', 'stack_h2_quad64', 'stack_h2_pair64')   # part forms: four workgroups per tile (stamps of part 0)
NTILE = 64 if DT == 'bf16' or F43 or H2 else 32
if PAIR:
    NTILE = 64 if net.last_path().endswith('64') else 32
if H2:   # the split-fp16 launch picks 32-frame tiles when the 64-frame ones would fill at most half of the CUs (stack_rows, diffnet.hip)
    nct = int(os.environ.get('BSG_H2_NCT', '0')) or (2 if B * ((T + 63) // 64) * 2 > 256 else 1)
    NTILE = 32 * nct
tiles, L = B * ((T + NTILE - 1) // NTILE), 20
st = torch.zeros(tiles * L * 8, dtype=torch.int64, device=dev)
for rep in range(3):
    st.zero_()
    # the stamped launch right behind ~1.5 s of back-to-back launches of the same form: the clock the chip holds under the launch needs seconds
    # to settle (MI355X_MICROARCH.md 'DVFS give-back' item 6), and a launch stamped from idle runs at another one
    for _ in range(int(os.environ.get('STAMP_WARM', '500'))):
        net(x, t, cond)
    _lib.check(_lib.load().bsg_diffnet_debug_stack_stamps(net._h, 50, B, T, _lib.ptr(st), _lib.stream_ptr()), 'stamps')
    torch.cuda.synchronize()
CYC = os.environ.get('BSG_STAMP_MODE') in ('2', '3') and H2
M3 = os.environ.get('BSG_STAMP_MODE') == '3' and net.last_path() == 'stack_h2q'   # slots 1 / 2 = conditioner loads issued / image rows written      # stamps carry the shader-cycle counter in their upper half: phases in cycles too
raw_all = st.cpu().numpy().reshape(tiles, L, 8)
if CYC:
    cy = ((raw_all >> 32) & 0xffffffff).astype(np.float64)
    raw_real = (raw_all & 0xffffffff).astype(np.float64)
    raw_real[:, L - 1, 6:] = 0
    s = raw_real / 100.0
else:
    s = raw_all.astype(np.float64) / 100.0      # us
names = ['GEMM1 (A->1)', 'gate+z+barrier (1->2)', 'GEMM2 residual (2->3)' if DT != 'bf16' else 'GEMM2 (2->3)',
         'publish+GEMM2 skip (3->4)' if DT != 'bf16' else 'core image + barrier C1 (3->4)', 'drain+barrier C (4->5)',
         'flag wait (5->6)', 'acquire+barrier D (6->7)', 'halo load+barrier A (7->0 next)']
if DT == 'bf16' or H2:   # stamps in time order: 0 start, 1 flags seen, 2 halo rows in place, 3 GEMM1 done, 4 gate done, 5 GEMM2 done, 6 image, 7 flag stored
    names = ['centre tap + flag wait (0->1)', 'halo copy (1->2)', 'outer taps (2->3)', 'gate+z+barrier (3->4)', 'GEMM2 + x/skip update (4->5)',
             'cond request + image + barrier C1 (5->6)', 'publish + drain + barrier C (6->7)', 'loop top (7->0 next)']
if F43:   # 0 start, 1 GEMM1 done (wave 0), 2 barrier B passed, 3 residual rows done, 4 image + publish + flag, 5 skip rows stored, 6 flags seen, 7 halo copied
    names = ['GEMM1, wave 0 (0->1)', 'gate + x recovery + barrier B (1->2)', 'GEMM2 residual rows (2->3)', 'image + barrier + publish + drain + flag (3->4)',
             'GEMM2 skip rows + skip RMW (4->5)', 'd tables + flag wait + barrier D (5->6)', 'halo copy (6->7)', 'barrier A (7->0 next)']
if PAIR:   # 0 conditioner term arrived (loop top), 1 flags seen, 2 image complete, 3 GEMM1 done, 4 partner's z published, 5 z complete, 6 GEMM2 + x update, 7 image flag stored
    names = ['own centre tap + wait for the image flags (0->1)', "partners' parts + halo copy (1->2)", 'rest of GEMM1 + conditioner term (2->3)',
             "gate + z out + wait for the partners' z (3->4)", 'z in + barrier (4->5)', 'GEMM2 + x/skip update (5->6)',
             'image + out + drain + flag (6->7)', 'loop top (7->0 next)']
if F43 or H2:
    raw = st.cpu().numpy().reshape(tiles, L, 8).astype(np.float64)
    cyc = raw[:, L - 1, 7] - raw[:, L - 1, 6]
    us = (raw[:, L - 1, 5] - raw[:, 0, 0]) / 100.0 if not CYC else (raw_real[:, L - 1, 5] - raw_real[:, 0, 0]) / 100.0
    print(f'shader clock held over the launch: {np.median(cyc / us):.0f} MHz (s_memtime cycles / s_memrealtime span, median over tiles)')
    tpr_ = (T + NTILE - 1) // NTILE
    print('  per-row shader clock (MHz):', ' '.join(f'{v:.0f}' for v in (cyc / us).reshape(B, tpr_).mean(1)))
inner = s[:, 1:L - 1]                                                          # layers with all 8 stamps and a successor
ORDER = [0, 3, 4, 5, 1, 2, 6, 7] if M3 else list(range(8))
if M3:
    names = ['GEMM1 (0->3)', 'gate+z+barrier (3->4)', 'GEMM2 + x/skip update (4->5)', 'conditioner loads issued (5->1)', 'image rows written (1->2)',
             'barrier C1 (2->6)', 'publish + drain + barrier C (6->7)', 'loop top (7->0 next)']
d = [inner[:, :, ORDER[i + 1]] - inner[:, :, ORDER[i]] for i in range(7)] + [s[:, 2:L, 0] - inner[:, :, 7]]
period = s[:, 2:L, 0] - s[:, 1:L - 1, 0]
print(f'path {net.last_path()}  B={B} T={T}: {tiles} tiles; layer period {period.mean():.1f} us (min {period.min():.1f}, max {period.max():.1f}); '
      f'kernel span {(s[:, L - 1, 4].max() - s[:, 0, 0].min()):.0f} us')
if CYC:
    ci = cy[:, 1:L - 1]
    dc = [(ci[:, :, ORDER[i + 1]] - ci[:, :, ORDER[i]]) % 2.0 ** 32 for i in range(7)] + [(cy[:, 2:L, 0] - ci[:, :, 7]) % 2.0 ** 32]
    pc = (cy[:, 2:L, 0] - cy[:, 1:L - 1, 0]) % 2.0 ** 32
    print(f'  layer period in shader cycles: {pc.mean():.0f}  ({pc.mean() / period.mean():.0f} MHz over the inner layers; MFMA cycles per SIMD and layer: 49152)')
for k, (n, v) in enumerate(zip(names, d)):
    extra = f'   {dc[k].mean():8.0f} cycles  {dc[k].mean() / v.mean():6.0f} MHz' if CYC else ''
    print(f'  {n:34s} mean {v.mean():7.2f} us   p10 {np.percentile(v, 10):7.2f}   p90 {np.percentile(v, 90):7.2f}{extra}')
a10 = s[:, 10, 0] - s[:, 10, 0].min()
print(f'  start of layer 10 over tiles: spread {a10.max():.1f} us, std {a10.std():.1f}; first half of the grid {a10[:tiles // 2].mean():.1f}, '
      f'second half {a10[tiles // 2:].mean():.1f}')
tpr = (T + NTILE - 1) // NTILE
rows = s[:, 10, 0].reshape(B, tpr) - s[:, 10, 0].min()
print('  per-row start of layer 10 (us):', ' '.join(f'{v:.0f}' for v in rows.mean(1)))
print('  per-row end of last layer (us): ', ' '.join(f'{v:.0f}' for v in (s[:, L - 1, 4].reshape(B, tpr) - s[:, 0, 0].min()).mean(1)))
print('  per-row spread inside a row at layer 10 (us):', ' '.join(f'{v:.0f}' for v in (rows.max(1) - rows.min(1))))
per = (s[:, 2:L, 0] - s[:, 1:L - 1, 0]).reshape(B, tpr, -1).mean((1, 2))
print('  per-row mean layer period (us):', ' '.join(f'{v:.0f}' for v in per))
if H2 or PAIR:
    # per-row phase means (rows 2 r, 2 r + 1 of the B = 16 launch run on XCD r): which phase makes the rows of some XCDs slower?
    print('  per-row phase means (us):')
    for n, v in zip(names, d):
        vr = v.reshape(B, tpr, -1).mean((1, 2))
        print(f'    {n[:44]:44s} ' + ' '.join(f'{x:5.2f}' for x in vr))
