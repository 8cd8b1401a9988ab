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


def h2_stress(net, x, cond, mode):
    """Operand distributions at the edges of the split-fp16 scheme (csrc/diffnet_h2.hip), applied identically by the parent (float64
    oracle) and the children (HIP forms) of tests/test_gpu_h2.py.  Returns (x, cond) (numpy, possibly rescaled); edits ``net`` in place.
      'outlier_w'  one weight per GEMM 10^3 x the largest of its layer: the per-layer power-of-two scale is chosen from max |w|, so
                   every other weight of the layer sits 2^10 lower in the fp16 range (their lo terms lose relative precision)
      'big_act'    diffusion-projection biases of +-2e4 on a third of the channels: |x + d| of 10^2 .. 5e4, below the 60 000 guard,
                   where the hi term has an ulp of 16 .. 32 and the lo term carries the rest
      'tiny_rows'  batch rows whose x and condition are 1e-6 x the others': activations whose lo terms are subnormal fp16"""
    with torch.no_grad():
        if mode == 'outlier_w':
            for i, layer in enumerate(net.residual_layers):
                w = layer.dilated_conv.weight
                w[(7 * i + 3) % w.shape[0], (11 * i + 5) % w.shape[1], i % 3] = 1000.0 * float(w.detach().abs().max())
                o = layer.output_projection.weight      # the outlier sits in a SKIP row (the residual rows feed back through 20 layers:
                C = o.shape[0] // 2                     # an outlier there makes the network itself ill-conditioned, for every arithmetic)
                o[C + (13 * i + 1) % C, (5 * i + 2) % o.shape[1], 0] = -1000.0 * float(o.detach().abs().max())
        elif mode == 'big_act':
            for i, layer in enumerate(net.residual_layers):
                b = layer.diffusion_projection.bias
                sel = torch.arange(b.numel()) % 3 == i % 3
                b[sel] += torch.where(torch.arange(b.numel())[sel] % 2 == 0, 2.0e4, -2.0e4) * (0.005 + 0.5 * ((i * 7) % 5))
        elif mode == 'tiny_rows':
            x, cond = x.copy(), cond.copy()
            x[::2] *= np.float32(1e-6)
            cond[::2] *= np.float32(1e-6)
        elif mode:
            raise ValueError(mode)
    return x, cond
