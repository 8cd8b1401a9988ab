#!/usr/bin/env python3
"""Time the parts of a mel-generation pass that are NOT the sampler loop: FS2-MIDI encoder + decoder and the hoisted conditioner
projections (bsg_diffnet_prepare), at the bench shapes (T = 1000).  BSG_GEMM_SPLIT=0 selects the fp32-matrix-pipe GEMM for an A/B.
    python tools/bench_fs2.py [B ...]          (default 16 64)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
model = bench.build_model(dev)
if os.environ.get('BSG_DTYPE') == 'bf16':
    model.denoise_fn.set_compute('bf16')
out = {}
for B in [int(v) for v in (sys.argv[1:] or ['16', '64'])]:
    wl = bench.Workload(model, dev, B, 0, 1)
    d = wl.d

    def fs2():
        return model.fs2(d['txt_tokens'], d['mel2ph'], d['spk_embed'], None, None, None, None, skip_decoder=False, infer=True, **wl.kw)

    ret = fs2()
    cond = ret['decoder_inp'].transpose(1, 2).contiguous()
    model.denoise_fn.prepare(cond)
    torch.cuda.synchronize()
    res = {}
    for name, fn in (('fs2_ms', fs2), ('prepare_ms', lambda: model.denoise_fn.prepare(cond))):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = round(e0.elapsed_time(e1) / 5, 3)
    out[B] = res
print(json.dumps(out))
