"""GPU parity, SURVEY.md §8 row f2: PitchExtractor and NSF-HiFiGAN against goldens produced by the reference."""
from collections import OrderedDict

import numpy as np
import pytest
import torch
import yaml

from bisinger_amd import synth
from tests.util import ROOT, maxabs, use_config

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def test_pitch_extractor_golden(gold, sd_spec):
    hp = use_config()
    hp.update(pitch_type='frame', use_uv=True, pitch_norm='log')
    from bisinger_amd.pe import PitchExtractor
    pe = PitchExtractor()
    mine = [[k, list(v.shape)] for k, v in pe.state_dict().items()]
    assert mine == sd_spec['PitchExtractor']
    spec = OrderedDict((k, tuple(s)) for k, s in sd_spec['PitchExtractor'])
    w = synth.synth_state_dict(spec, seed=11)
    for k in spec:
        if k.endswith('running_var'):
            w[k] = (0.5 + np.abs(w[k]) * 5).astype(np.float32)
        if k.endswith('running_mean'):
            w[k] = (w[k] * 3).astype(np.float32)
    pe.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    pe = pe.cuda()
    g = gold('f2')
    rs = np.random.RandomState(31)
    for tag, (B, T) in {'B2T50': (2, 50), 'B1T133': (1, 133)}.items():
        mel = (rs.standard_normal((B, T, 80)) * 1.5 - 3.0).astype(np.float32)
        if B > 1:
            mel[1, T - 7:] = 0
        r = pe(torch.from_numpy(mel).cuda())
        assert maxabs(r['pitch_pred'], g[f'pe.{tag}.pitch_pred']) <= 1e-4
        want = g[f'pe.{tag}.f0']
        got = r['f0_denorm_pred'].cpu().numpy()
        # the voiced/unvoiced decision is a sign test on a network output: exclude frames within rounding of 0
        near = np.abs(g[f'pe.{tag}.pitch_pred'][:, :, 1]) < 1e-4
        assert np.abs(got - want)[~near].max() <= 2e-4 * max(1.0, np.abs(want).max())
        assert (got[1, T - 7:] == 0).all() if B > 1 else True


def test_nsf_hifigan_golden(gold, sd_spec):
    from bisinger_amd.hifigan import HifiGanGenerator
    cfg = yaml.safe_load(open(f'{ROOT}/bisinger_amd/configs/hifigan.yaml'))
    cfg['use_pitch_embed'] = True
    gen = HifiGanGenerator(cfg)
    mine = [[k, list(v.shape)] for k, v in gen.state_dict().items()]
    assert mine == sd_spec['HifiGanGenerator_nsf_weight_norm']
    spec = OrderedDict((k, tuple(s)) for k, s in sd_spec['HifiGanGenerator_nsf_weight_norm'])
    gen.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(spec, seed=13).items()}, strict=True)
    gen = gen.cuda()
    gen.remove_weight_norm()
    g = gold('f2')
    rs = np.random.RandomState(31)
    for B, T in ((2, 50), (1, 133)):            # replay the generator script's draws to stay in sync
        rs.standard_normal((B, T, 80))
    for tag, (B, T) in {'B1T12': (1, 12), 'B2T21': (2, 21)}.items():
        mel = (rs.standard_normal((B, 80, T)) * 1.5 - 3.0).astype(np.float32)
        f0 = (rs.uniform(90, 500, size=(B, T))).astype(np.float32)
        f0[:, T // 3: T // 3 + 3] = 0
        rand_ini = rs.uniform(0, 1, size=(B, 9)).astype(np.float32)
        noise = rs.standard_normal((B, T * 256, 9)).astype(np.float32)
        y = gen(torch.from_numpy(mel).cuda(), torch.from_numpy(f0).cuda(), rand_ini=torch.from_numpy(rand_ini), noise=torch.from_numpy(noise))
        assert y.shape == (B, 1, T * 256)
        assert maxabs(y, g[f'nsf.{tag}.wav']) <= 2e-4, tag
    # generated draws: runs, deterministic per seed
    a = gen(torch.from_numpy(mel).cuda(), torch.from_numpy(f0).cuda(), seed=3)
    b = gen(torch.from_numpy(mel).cuda(), torch.from_numpy(f0).cuda(), seed=3)
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())
