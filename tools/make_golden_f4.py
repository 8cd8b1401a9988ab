#!/usr/bin/env python3
"""Golden for SURVEY.md §8 row f4 (FFT candidate denoiser) from the reference's own class.  Build container only."""
import json
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from bisinger_amd import synth          # noqa: E402
import ref_import                       # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
torch.set_grad_enabled(False)
R = ref_import.import_reference()
hp = R['hparams']
from usr.diff.candidate_decoder import FFT      # noqa: E402
from oracle import candidate_decoder as ocd      # noqa: E402
m = FFT(hp['hidden_size'], hp['dec_layers'], hp['dec_ffn_kernel_size'], hp['num_heads']).eval()
spec = OrderedDict((k, tuple(v.shape)) for k, v in m.state_dict().items())
w = synth.synth_state_dict(spec, seed=17)
m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
sd = {k: v.clone() for k, v in m.state_dict().items()}
rs = np.random.RandomState(41)
out, rep = {}, {}
for tag, (B, T) in {'B2T40': (2, 40), 'B1T77': (1, 77)}.items():
    x = rs.standard_normal((B, 1, 80, T)).astype(np.float32)
    cond = rs.standard_normal((B, 256, T)).astype(np.float32)
    t = rs.randint(0, 100, size=(B,)).astype(np.int64)
    y = m(torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(cond))
    out[f'{tag}.eps'] = y.numpy()
    mine = ocd.fft_denoiser_forward(sd, torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(cond))
    rep[f'fft_denoiser.{tag}'] = float((mine - y).abs().max())
np.savez_compressed(os.path.join(GOLD, 'f4.npz'), **out)
js = json.load(open(os.path.join(GOLD, 'state_dict_spec.json')))
js['FFT'] = [[k, list(s)] for k, s in spec.items()]
json.dump(js, open(os.path.join(GOLD, 'state_dict_spec.json'), 'w'), indent=0)
r0 = json.load(open(os.path.join(GOLD, 'oracle_vs_reference.json')))
r0.update(rep)
json.dump(r0, open(os.path.join(GOLD, 'oracle_vs_reference.json'), 'w'), indent=1)
print(rep, len(spec))
