#!/usr/bin/env python3
"""Goldens for SURVEY.md §8 row f2 (PitchExtractor + NSF-HiFiGAN) from the reference's own modules.
Build container only (needs /root/reference).  Writes tests/golden/f2.npz and extends state_dict_spec.json."""
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


def main():
    R = ref_import.import_reference()
    hp = R['hparams']
    hp['pitch_type'], hp['use_uv'], hp['pitch_norm'] = 'frame', True, 'log'
    from modules.fastspeech.pe import PitchExtractor
    from oracle import nsf as onsf, pe as ope
    out, report = {}, {}
    # ---- PitchExtractor
    pe = PitchExtractor().eval()
    spec = OrderedDict((k, tuple(v.shape)) for k, v in pe.state_dict().items())
    w = synth.synth_state_dict(spec, seed=11)
    for k in spec:                          # BatchNorm running statistics: positive variances
        if k.endswith('running_var'):
            w[k] = (0.5 + np.abs(w[k]) * 5).astype(np.float32)
        if k.endswith('running_mean'):
            w[k] = (w[k] * 3).astype(np.float32)
    missing, unexpected = pe.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    assert not unexpected and all(synth.is_computed_buffer(k) or k.endswith('num_batches_tracked') for k in missing), missing
    sd = {k: v.clone() for k, v in pe.state_dict().items()}
    rs = np.random.RandomState(31)
    for tag, (B, T) in {'B2T50': (2, 50), 'B1T133': (1, 133)}.items():
        mel = (rs.standard_normal((B, T, 80)) * 1.5 - 3.0).astype(np.float32)
        if B > 1:
            mel[1, T - 7:] = 0          # padded frames
        r = pe(torch.from_numpy(mel))
        out[f'pe.{tag}.pitch_pred'] = r['pitch_pred'].numpy()
        out[f'pe.{tag}.f0'] = r['f0_denorm_pred'].numpy()
        mine = ope.pitch_extractor_forward(sd, torch.from_numpy(mel))
        report[f'pe.{tag}.pitch_pred'] = float((mine['pitch_pred'] - r['pitch_pred']).abs().max())
        report[f'pe.{tag}.f0'] = float((mine['f0_denorm_pred'] - r['f0_denorm_pred']).abs().max())
    # ---- NSF-HiFiGAN
    Gen, cfg = ref_import.import_hifigan()
    cfg = dict(cfg)
    cfg['use_pitch_embed'] = True
    g = Gen(cfg)
    hspec = OrderedDict((k, tuple(v.shape)) for k, v in g.state_dict().items())
    hw = synth.synth_state_dict(hspec, seed=13)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in hw.items()}, strict=True)
    hsd = {k: v.clone() for k, v in g.state_dict().items()}
    g.remove_weight_norm(); g.eval()
    import modules.parallel_wavegan.models.source as src
    for tag, (B, T) in {'B1T12': (1, 12), 'B2T21': (2, 21)}.items():
        mel = (rs.standard_normal((B, 80, T)) * 1.5 - 3.0).astype(np.float32)
        f0 = (rs.uniform(90, 500, size=(B, T))).astype(np.float32)
        f0[:, T // 3: T // 3 + 3] = 0       # an unvoiced stretch
        L = T * 256
        rand_ini = rs.uniform(0, 1, size=(B, 9)).astype(np.float32)
        noise = rs.standard_normal((B, L, 9)).astype(np.float32)
        o_rand, o_randn_like = torch.rand, torch.randn_like
        calls = {'n': 0}

        def fake_randn_like(x, **k):
            calls['n'] += 1
            return torch.from_numpy(noise) if tuple(x.shape) == noise.shape else o_randn_like(x, **k)
        torch.rand = lambda *a, **k: torch.from_numpy(rand_ini).clone()
        torch.randn_like = fake_randn_like
        try:
            y = g(torch.from_numpy(mel), torch.from_numpy(f0))
        finally:
            torch.rand, torch.randn_like = o_rand, o_randn_like
        out[f'nsf.{tag}.wav'] = y.numpy()
        mine = onsf.nsf_hifigan_forward(hsd, torch.from_numpy(mel), torch.from_numpy(f0), torch.from_numpy(rand_ini),
                                        torch.from_numpy(noise), cfg)
        report[f'nsf.{tag}.wav'] = float((mine - y).abs().max())
        har = g.m_source
        torch.rand = lambda *a, **k: torch.from_numpy(rand_ini).clone()
        torch.randn_like = fake_randn_like
        try:
            f0u = g.f0_upsamp(torch.from_numpy(f0)[:, None]).transpose(1, 2)
            hs, _, _ = har(f0u)
        finally:
            torch.rand, torch.randn_like = o_rand, o_randn_like
        out[f'nsf.{tag}.har'] = hs.transpose(1, 2).numpy()
        mine_h = onsf.sine_source(hsd, torch.from_numpy(f0), torch.from_numpy(rand_ini), torch.from_numpy(noise), cfg['audio_sample_rate'], 256)
        report[f'nsf.{tag}.har'] = float((mine_h - hs.transpose(1, 2)).abs().max())
    np.savez_compressed(os.path.join(GOLD, 'f2.npz'), **out)
    js = json.load(open(os.path.join(GOLD, 'state_dict_spec.json')))
    js['PitchExtractor'] = [[k, list(s)] for k, s in spec.items()]
    js['HifiGanGenerator_nsf_weight_norm'] = [[k, list(s)] for k, s in hspec.items()]
    json.dump(js, open(os.path.join(GOLD, 'state_dict_spec.json'), 'w'), indent=0)
    rep = json.load(open(os.path.join(GOLD, 'oracle_vs_reference.json')))
    rep.update(report)
    json.dump(rep, open(os.path.join(GOLD, 'oracle_vs_reference.json'), 'w'), indent=1)
    for k, v in report.items():
        print(f'  {k:32s} {v}')
    print('size KB', os.path.getsize(os.path.join(GOLD, 'f2.npz')) / 1024)


if __name__ == '__main__':
    main()
