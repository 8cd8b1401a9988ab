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
