#!/usr/bin/env python3
"""Micro-benchmark of the shared fp32 MFMA GEMM (bsg_gemm_f32) on the shapes the path uses (GPU box)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bisinger_amd import _lib  # noqa: E402

lib = _lib.load()
SHAPES = [  # name, M, N, K, trans_b, batch
    ('QKV proj      [B*T,256]x[768,256]^T', 16000, 768, 256, 1, 1),
    ('FFN linear    [B*T,1024]x[256,1024]^T', 16000, 256, 1024, 1, 1),
    ('out proj      [B*T,256]x[256,256]^T', 16000, 256, 256, 1, 1),
    ('QK^T          32 x [1000,128]x[1000,128]^T', 1000, 1000, 128, 1, 32),
    ('PV            32 x [1000,1000]x[1000,128]', 1000, 128, 1000, 0, 32),
    ('cond 1x1 conv 16 x [512,256]x[256,1000]', 512, 1000, 256, 0, 16),
    ('big square    [4096,4096]x[4096,4096]^T', 4096, 4096, 4096, 1, 1),
]
for name, M, N, K, tb, batch in SHAPES:
    a = torch.randn(batch, M, K, device='cuda')
    b = torch.randn(batch, N, K, device='cuda') if tb else torch.randn(batch, K, N, device='cuda')
    c = torch.empty(batch, M, N, device='cuda')
    run = lambda: _lib.check(lib.bsg_gemm_f32(_lib.ptr(a), _lib.ptr(b), _lib.ptr(c), None, None, M, N, K, K, K if tb else N, N, tb, batch,
                                              M * K, N * K, M * N, 0, _lib.stream_ptr()), 'gemm')
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'{name:48s} {ms * 1e3:8.1f} us  {2.0 * M * N * K * batch / ms / 1e9:6.1f} TFLOP/s', flush=True)

# the pre-split GEMM (gemm_h2w.hip) on the same products: `reps` launches inside one call (the call also packs the weights and splits the
# activation once: timed with reps = 1 and reps = 21, the difference / 20 is one launch)
import time  # noqa: E402
H2W = [  # name, rows, Wn, K, taps, act_is_a, batch
    ('h2w QKV proj      [16000,256] x [768,256]^T', 16000, 768, 256, 1, 1, 1),
    ('h2w out proj      [16000,256] x [256,256]^T', 16000, 256, 256, 1, 1, 1),
    ('h2w FFN conv k=9  16 x [1000,256] * [9][1024,256]', 1000, 1024, 256, 9, 1, 16),
    ('h2w FFN linear    [16000,1024] x [256,1024]^T', 16000, 256, 1024, 1, 1, 1),
    ('h2w cond 1x1      320 x [512,256] x [1000,256]^T', 1000, 512, 256, 1, 0, 320),
    ('h2w FFN conv k=9  8 x [1000,256] (rank shape)', 1000, 1024, 256, 9, 1, 8),
    ('h2w FFN conv k=9  1 x [1000,256] (B=1)', 1000, 1024, 256, 9, 1, 1),
    ('h2w QKV proj      [1000,256] (B=1)', 1000, 768, 256, 1, 1, 1),
]
for name, rows, Wn, K, taps, aia, batch in H2W:
    if name.startswith('h2w cond'):
        a = torch.randn(16, rows, K, device='cuda').repeat(20, 1, 1)   # 20 layers over 16 rows, as one batch
    else:
        a = torch.randn(batch, rows, K, device='cuda')
    w = torch.randn(taps, Wn, K, device='cuda') * 0.05
    c = torch.empty(batch, rows, Wn, device='cuda')

    def call(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(lib.bsg_gemm_presplit_f32(_lib.ptr(a), _lib.ptr(w), _lib.ptr(c), None, rows, Wn, K, taps, aia, batch, 0, reps, _lib.stream_ptr()), 'h2w')
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    call(3)
    t1 = min(call(1) for _ in range(3))
    t21 = min(call(21) for _ in range(3))
    us = (t21 - t1) / 20 * 1e6
    print(f'{name:52s} {us:8.1f} us  {2.0 * rows * Wn * K * taps * batch / us / 1e6:6.1f} TFLOP/s', flush=True)
