#!/usr/bin/env python3
"""Goldens for BASELINE.json's *stated* configurations, from the reference's own modules.  Build container only.

    python tools/make_golden_cfg.py       # writes tests/golden/cfg.npz, updates oracle_vs_reference.json

  cfg0     BASELINE configs[0] / SURVEY §8(d) "Config 1": B=1, T_txt=50, T=500, FS2-MIDI encoder + decoder +
           100-step DDPM decoder (linear beta to 0.06), supplied noise.  Reference entry:
           train_bisinger/usr/diff/shallow_diffusion_tts.py:230-273.  Stored: mel_out [1,500,80] (160 KB).
  shipped  the configuration every shipped BiSinger experiment runs: timesteps = K_step = 1000, max_beta 0.02,
           pndm_speedup 5 -> 200 PLMS iterations, 201 denoiser evaluations
           (usr/configs/lang-esm-style-ori-shift/diff.yaml:16-23, shallow_diffusion_tts.py:258-264), B=1 (the
           reference's PLMS branch is B=1-only, :189), T_txt=8, T=64.  Stored: mel_out [1,64,80].

Only outputs are stored; inputs, weights and noise are regenerated from bisinger_amd/synth.py formulas.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from bisinger_amd import synth          # noqa: E402
import ref_import                       # noqa: E402
from make_golden import SuppliedNoise, load_synth, sha, tin   # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
torch.set_grad_enabled(False)

from tests.cfg_fixtures import CFG0, SHIPPED, inputs_and_noise   # noqa: E402  (shapes, seeds, conditioning)


def main():
    R = ref_import.import_reference()
    hp, sdt = R['hparams'], R['sdt']
    from oracle import melgen as omg

    def build(timesteps, K_step, max_beta, gain=synth.DIFFNET_GAIN):
        hp['timesteps'], hp['K_step'], hp['max_beta'] = timesteps, K_step, max_beta
        betas = sdt.linear_beta_schedule(timesteps, max_beta=max_beta)   # the module captured max_beta at import (:44)
        m = R['GaussianDiffusion'](R['phone_encoder'], 80, R['DiffNet'](80), timesteps=timesteps, K_step=K_step,
                                   loss_type='l1', betas=betas, spec_min=hp['spec_min'], spec_max=hp['spec_max']).eval()
        load_synth(m, 0, gain)
        return m

    out, rep = {}, {}
    # ---- configs[0] --------------------------------------------------------------------------------
    c = CFG0
    model = build(c['timesteps'], c['K_step'], c['max_beta'], c['gain'])
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    inp, noise = inputs_and_noise(c)
    ti = tin(inp)
    kw = {k: ti[k] for k in ('pitch_midi', 'midi_dur', 'is_slur', 'lang', 'speechsing')}
    with SuppliedNoise(sdt, noise):
        ref = model(ti['txt_tokens'], mel2ph=ti['mel2ph'], spk_embed=ti['spk_embed'], ref_mels=None, infer=True, **kw)
    out['cfg0.mel_out'] = ref['mel_out'].numpy()
    out['cfg0.sha_in'] = np.array(sha(*[inp[k] for k in sorted(inp)], noise))
    mine = omg.mel_gen(sd, ti, torch.from_numpy(noise))
    rep['cfg0.mel_out'] = float((mine['mel_out'] - ref['mel_out']).abs().max())
    print('cfg0: oracle vs reference', rep['cfg0.mel_out'])

    # ---- shipped configuration (1000-step schedule, PLMS interval 5) ----------------------------------
    c = SHIPPED
    model = build(c['timesteps'], c['K_step'], c['max_beta'], c['gain'])
    hp['pndm_speedup'] = c['pndm_speedup']
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    inp, noise = inputs_and_noise(c)          # PLMS is deterministic after x_T
    ti = tin(inp)
    kw = {k: ti[k] for k in ('pitch_midi', 'midi_dur', 'is_slur', 'lang', 'speechsing')}
    with SuppliedNoise(sdt, noise):
        ref = model(ti['txt_tokens'], mel2ph=ti['mel2ph'], spk_embed=ti['spk_embed'], ref_mels=None, infer=True, **kw)
    hp['pndm_speedup'] = 0
    out['shipped.mel_out'] = ref['mel_out'].numpy()
    out['shipped.sha_in'] = np.array(sha(*[inp[k] for k in sorted(inp)], noise))
    mine = omg.mel_gen(sd, ti, torch.from_numpy(noise), timesteps=1000, K_step=1000, max_beta=0.02, pndm_speedup=5)
    rep['shipped.mel_out'] = float((mine['mel_out'] - ref['mel_out']).abs().max())
    print('shipped: oracle vs reference', rep['shipped.mel_out'])

    # ---- length bucketing: the reference's batch_by_size on size-ordered indices (utils/__init__.py:90-143) -----------
    from utils import batch_by_size
    cases = []
    rs = np.random.RandomState(5)
    for n, max_tokens, max_sentences in ((12, 4000, None), (40, 6000, 8), (25, 3000, 3), (7, None, 2), (30, 36000, 28)):
        lens = [int(v) for v in rs.randint(50, 1500, size=n)]
        order = sorted(range(n), key=lambda i: (-lens[i], i))
        ref_b = batch_by_size(order, lambda i: lens[i], max_tokens=max_tokens, max_sentences=max_sentences)
        cases.append({'lengths': lens, 'max_frames': max_tokens, 'max_sentences': max_sentences,
                      'batches': [[int(i) for i in b] for b in ref_b]})
    json.dump(cases, open(os.path.join(GOLD, 'buckets.json'), 'w'))
    print(f'wrote tests/golden/buckets.json ({len(cases)} cases)')

    path = os.path.join(GOLD, 'cfg.npz')
    np.savez_compressed(path, **out)
    print(f'wrote {os.path.relpath(path, ROOT)} ({os.path.getsize(path) / 1024:.1f} KB)')
    rp = os.path.join(GOLD, 'oracle_vs_reference.json')
    r0 = json.load(open(rp))
    r0.update(rep)
    json.dump(r0, open(rp, 'w'), indent=1)


if __name__ == '__main__':
    main()
