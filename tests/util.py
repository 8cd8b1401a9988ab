"""Helpers shared by the GPU parity tests: formula weights loaded into the drop-in modules."""
import json
import os
from collections import OrderedDict

import numpy as np
import torch

from bisinger_amd import synth
from bisinger_amd.hparams import hparams, set_hparams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG100 = os.path.join(ROOT, 'bisinger_amd', 'configs', 'bisinger_diff100.yaml')


def use_config(hparams_str=''):
    set_hparams(CFG100, print_hparams=False, hparams_str=hparams_str)
    return hparams


def load_formula_weights(module, seed=0, gain=None, prefix=''):
    """Load synth weights keyed by ``prefix + state_dict key`` into ``module`` (strict on non-buffers)."""
    sd = module.state_dict()
    spec = OrderedDict((prefix + k, tuple(v.shape)) for k, v in sd.items())
    w = synth.synth_state_dict(spec, seed, gain)
    new = {k[len(prefix):]: torch.from_numpy(v) for k, v in w.items()}
    missing, unexpected = module.load_state_dict(new, strict=False)
    assert not unexpected, unexpected
    assert all(synth.is_computed_buffer(k) for k in missing), missing
    return module


def cpu_sd(module, prefix=''):
    return {prefix + k: v.detach().cpu() for k, v in module.state_dict().items()}


def maxabs(a, b):
    a = a.detach().cpu().double() if torch.is_tensor(a) else torch.from_numpy(np.asarray(a)).double()
    b = b.detach().cpu().double() if torch.is_tensor(b) else torch.from_numpy(np.asarray(b)).double()
    return float((a - b).abs().max())
