import json
import os
import sys
from collections import OrderedDict

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the CPU oracle is the slow side of every parity test; PyTorch's default of one thread per hardware thread is far
    # from the fastest setting on a many-core host (bench.py calibrates the same way)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope='session')
def gold():
    return lambda name: np.load(os.path.join(GOLD, name + '.npz'))


@pytest.fixture(scope='session')
def sd_spec():
    with open(os.path.join(GOLD, 'state_dict_spec.json')) as f:
        js = json.load(f)
    return js


@pytest.fixture(scope='session')
def gd_sd(sd_spec):
    """Formula weights + computed buffers of the 100-step GaussianDiffusion, as torch CPU tensors."""
    from bisinger_amd import synth
    from oracle import diffusion as odf
    spec = OrderedDict((k, tuple(s)) for k, s in sd_spec['GaussianDiffusion'])
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(spec, 0, synth.DIFFNET_GAIN).items()}
    sd.update(odf.make_schedule(100, 'linear', 0.06))
    g = np.load(os.path.join(GOLD, 'schedules.npz'))
    sd['spec_min'] = torch.from_numpy(g['spec_min'])
    sd['spec_max'] = torch.from_numpy(g['spec_max'])
    return sd


@pytest.fixture(scope='session')
def hifigan_sd(sd_spec):
    from bisinger_amd import synth
    spec = OrderedDict((k, tuple(s)) for k, s in sd_spec['HifiGanGenerator_weight_norm'])
    return {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(spec, 7).items()}
