#!/usr/bin/env python3
"""Profiling driver: N HiFi-GAN forwards at B=16, T=1000 (run under rocprofv3 --kernel-trace --stats on the GPU box)."""
import os
import sys
from collections import OrderedDict

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bisinger_amd import synth  # noqa: E402
from bisinger_amd.hifigan import HifiGanGenerator  # noqa: E402

torch.set_grad_enabled(False)
cfg = yaml.safe_load(open(f'{ROOT}/bisinger_amd/configs/hifigan.yaml'))
voc = HifiGanGenerator(cfg)
spec = OrderedDict((k, tuple(v.shape)) for k, v in voc.state_dict().items())
voc.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(spec, 7).items()})
voc = voc.cuda()
voc.remove_weight_norm()
B, T = int(os.environ.get('PB', 16)), int(os.environ.get('PT', 1000))
mel = torch.randn(B, 80, T, device='cuda')
for _ in range(int(os.environ.get("PN", 5))):
    wav = voc(mel)
torch.cuda.synchronize()
print('done', tuple(wav.shape))
