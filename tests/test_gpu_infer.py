"""GPU: the inference entry points (DiffSingerE2EInfer) on a synthetic checkpoint directory laid out like the
reference's (checkpoints/<exp>/model_ckpt_steps_N.ckpt with 'model.' keys, vocoder dir with config.yaml +
['state_dict']['model_gen'], binary_data_dir with phone_set.json / spk_map.json) — end to end against the oracle."""
import json
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch
import yaml

from bisinger_amd import synth
from tests.util import ROOT, maxabs

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture()
def workdir(tmp_path, sd_spec, gd_sd, hifigan_sd, monkeypatch):
    monkeypatch.chdir(tmp_path)
    os.makedirs('checkpoints/exp_diff_e2e')
    os.makedirs('checkpoints/hifigan')
    os.makedirs('data/binary')
    full = dict(gd_sd)
    full['fs2.decoder.embed_positions._float_tensor'] = torch.zeros(1)
    torch.save({'state_dict': {'model.' + k: v for k, v in full.items()}, 'global_step': 1000}, 'checkpoints/exp_diff_e2e/model_ckpt_steps_1000.ckpt')
    torch.save({'state_dict': {'model.' + k: v * 0 for k, v in full.items()}}, 'checkpoints/exp_diff_e2e/model_ckpt_steps_200.ckpt')
    torch.save({'state_dict': {'model_gen': hifigan_sd}}, 'checkpoints/hifigan/model_ckpt_steps_5.ckpt')
    hcfg = yaml.safe_load(open(f'{ROOT}/bisinger_amd/configs/hifigan.yaml'))
    yaml.safe_dump(hcfg, open('checkpoints/hifigan/config.yaml', 'w'))
    json.dump(['<AP>', '<SP>'] + [f'p{i}' for i in range(60)], open('data/binary/phone_set.json', 'w'))
    json.dump({'Tenor-1': 3, 'Alto-2': 7}, open('data/binary/spk_map.json', 'w'))
    cfg = {'base_config': f'{ROOT}/bisinger_amd/configs/bisinger_diff100.yaml', 'binary_data_dir': 'data/binary',
           'vocoder_ckpt': 'checkpoints/hifigan', 'pe_enable': False, 'use_nsf': False, 'max_frames': 5000}
    yaml.safe_dump(cfg, open('exp.yaml', 'w'))
    return tmp_path


def _item(n, seed):
    rs = np.random.RandomState(seed)
    names = ['C4', 'D#4', 'Gb3', 'A4', 'rest', 'E4/E4']
    return {'input_type': 'phoneme', 'item_name': f'it{seed}', 'spk_name': 'Tenor-1' if seed % 2 else 'Alto-2', 'text': 'x',
            'ph_seq': ' '.join(['<AP>'] + [f'p{rs.randint(60)}' for _ in range(n - 1)]),
            'note_seq': ' '.join(names[rs.randint(len(names))] for _ in range(n)),
            'note_dur_seq': ' '.join(f'{rs.uniform(0.1, 0.6):.3f}' for _ in range(n)),
            'is_slur_seq': ' '.join(str(rs.randint(2)) for _ in range(n)),
            'lang_seq': ' '.join(str(rs.randint(2)) for _ in range(n))}


def _oracle_wav(infer, sample, seed, hifigan_sd, sd_spec):
    """Oracle pipeline with the same Philox noise stream as the device (bisinger_amd/synth.py)."""
    from oracle import fs2 as ofs2, hifigan as ohg, melgen as omg
    sd = {k: v.detach().cpu() for k, v in infer.model.state_dict().items()}
    inp = {'txt_tokens': sample['txt_tokens'].cpu(), 'spk_embed': sample['spk_ids'].cpu(), 'pitch_midi': sample['pitch_midi'].cpu(),
           'midi_dur': sample['midi_dur'].cpu(), 'is_slur': sample['is_slur'].cpu(), 'lang': sample['lang'].cpu(),
           'speechsing': sample['speechsing'].cpu()}
    f = ofs2.fs2_forward(sd, inp)
    B, T = f['mel2ph'].shape
    n = B * 80 * T
    noise = np.stack([synth.philox_normal(seed, 0, n)] + [synth.philox_normal(seed, i + 1, n) for i in reversed(range(100))])
    r = omg.mel_gen(sd, inp, torch.from_numpy(noise.reshape(101, B, 80, T)), fs2_out=f)
    return r, ohg.hifigan_forward(hifigan_sd, r['mel_out'].transpose(1, 2), sd_spec['hifigan_cfg'])


def test_e2e_infer_once_and_batch(workdir, hifigan_sd, sd_spec):
    from bisinger_amd.hparams import hparams, set_hparams
    from bisinger_amd.infer import DiffSingerE2EInfer, note_to_midi
    assert [note_to_midi(x) for x in ('C4', 'A4', 'F#3', 'Gb3', 'C-1')] == [60, 69, 54, 54, 0]
    set_hparams('exp.yaml', exp_name='exp_diff_e2e', print_hparams=False, hparams_str='seed=4321')
    assert hparams['work_dir'] == 'checkpoints/exp_diff_e2e' and hparams['timesteps'] == 100
    assert os.path.exists('checkpoints/exp_diff_e2e/config.yaml')      # saved like the reference (hparams.py:98-101)
    infer = DiffSingerE2EInfer(hparams)
    assert 'conv_pre.weight' in infer.vocoder.state_dict()             # weight norm removed
    inp = _item(9, 1)
    wav = infer.infer_once(inp)                                         # predicted durations, Philox noise seed 4321
    item = infer.preprocess_input(inp, 'phoneme')
    assert item['pitch_midi'].tolist()[0] in (60, 63, 54, 69, 0, 64)
    sample = infer.input_to_batch(item)
    r, want = _oracle_wav(infer, sample, 4321, hifigan_sd, sd_spec)
    assert wav.ndim == 1 and wav.shape[0] == r['mel_out'].shape[1] * 256
    assert maxabs(wav, want.reshape(-1)) <= 2e-3
    # batched extension: 3 items of different lengths
    items = [infer.preprocess_input(_item(n, s), 'phoneme') for n, s in ((9, 1), (6, 2), (11, 3))]
    wavs = infer.forward_batch(items, seed=77)
    sample = infer.collate(items)
    r, want = _oracle_wav(infer, sample, 77, hifigan_sd, sd_spec)
    for i, w in enumerate(wavs):
        n = int((r['mel2ph'][i] > 0).sum()) * 256
        assert w.shape == (n,)
        assert maxabs(w, want[i, 0, :n]) <= 2e-3
    # length-bucketed batches: 13 ragged items, budget on padded frames + rows per bucket; the unit of
    # reproducibility is the bucket (ESM attends over the batch axis), so the oracle runs per bucket
    specs = [(5, 11), (9, 12), (6, 13), (12, 14), (7, 15), (10, 16), (4, 17), (8, 18), (11, 19), (6, 20), (9, 21), (5, 22), (12, 23)]
    items = [infer.preprocess_input(_item(n, s), 'phoneme') for n, s in specs]
    est = [infer.estimate_frames(it) for it in items]
    assert all(e > 0 for e in est)
    budget = 3 * max(est)
    wavs = infer.forward_batch(items, seed=55, max_frames=budget, max_sentences=4)
    st = infer.last_batch_stats
    buckets = st['buckets']
    assert sorted(i for b in buckets for i in b) == list(range(len(items))) and len(buckets) >= 4
    assert all(len(b) <= 4 and len(b) * max(est[i] for i in b) <= budget for b in buckets)
    assert st['padded_frames'] >= st['real_frames'] == sum(st['frames']) and 0.0 <= st['waste'] < 1.0
    print(f"bucketing: {len(buckets)} buckets, padded-frame waste {st['waste']:.1%} vs {st['waste_single_batch']:.1%} as one batch "
          f"({st['padded_frames']} vs {st['padded_frames_single_batch']} padded frames for {st['real_frames']} real)")
    for b in (buckets[0], buckets[-1]):
        sample = infer.collate([items[i] for i in b])
        r, want = _oracle_wav(infer, sample, 55, hifigan_sd, sd_spec)
        for j, i in enumerate(b):
            n = int((r['mel2ph'][j] > 0).sum()) * 256
            assert wavs[i].shape == (n,) and st['frames'][i] * 256 == n
            assert maxabs(wavs[i], want[j, 0, :n]) <= 2e-3
    assert all(w is not None and np.isfinite(w).all() for w in wavs)
    with pytest.raises(NotImplementedError):
        infer.preprocess_input({'text': 'AP 你好 AP', 'notes': 'rest | C4 | rest', 'notes_duration': '0.1 | 0.2 | 0.1'}, 'word')


def test_e2e_with_pitch_extractor_and_nsf_vocoder(workdir, sd_spec):
    """pe_enable + use_nsf (the shipped M4Singer set-up, base.yaml:25,62): mel -> PitchExtractor f0 -> NSF-HiFiGAN."""
    import json as _json
    from collections import OrderedDict
    from bisinger_amd.hparams import hparams, set_hparams
    from bisinger_amd.infer import DiffSingerE2EInfer
    from oracle import fs2 as ofs2, melgen as omg, nsf as onsf, pe as ope
    os.makedirs('checkpoints/pe')
    os.makedirs('checkpoints/nsf')
    spec = OrderedDict((k, tuple(s)) for k, s in sd_spec['PitchExtractor'])
    pw = synth.synth_state_dict(spec, seed=11)
    for k in spec:
        if k.endswith('running_var'):
            pw[k] = (0.5 + np.abs(pw[k]) * 5).astype(np.float32)
    pe_sd = {k: torch.from_numpy(v) for k, v in pw.items()}
    for k, s_ in spec.items():
        if k not in pe_sd:
            pe_sd[k] = torch.zeros(s_, dtype=torch.long if k.endswith('num_batches_tracked') else torch.float32)
    torch.save({'state_dict': {'model.' + k: v for k, v in pe_sd.items()}}, 'checkpoints/pe/model_ckpt_steps_10.ckpt')
    nspec = OrderedDict((k, tuple(s)) for k, s in sd_spec['HifiGanGenerator_nsf_weight_norm'])
    nsd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(nspec, seed=13).items()}
    torch.save({'state_dict': {'model_gen': nsd}}, 'checkpoints/nsf/model_ckpt_steps_7.ckpt')
    hcfg = yaml.safe_load(open(f'{ROOT}/bisinger_amd/configs/hifigan.yaml'))
    hcfg['use_pitch_embed'] = True
    yaml.safe_dump(hcfg, open('checkpoints/nsf/config.yaml', 'w'))
    cfg = yaml.safe_load(open('exp.yaml'))
    cfg.update(vocoder_ckpt='checkpoints/nsf', pe_enable=True, pe_ckpt='checkpoints/pe', use_nsf=True, pitch_type='frame',
               use_uv=True, pitch_norm='log')
    yaml.safe_dump(cfg, open('exp2.yaml', 'w'))
    set_hparams('exp2.yaml', exp_name='exp_diff_e2e', print_hparams=False, hparams_str='seed=99')
    infer = DiffSingerE2EInfer(hparams)
    assert infer.vocoder.use_nsf and hasattr(infer, 'pe')
    inp = _item(8, 5)
    wav = infer.infer_once(inp)
    wav2 = infer.infer_once(inp)
    assert wav.ndim == 1 and np.isfinite(wav).all() and np.array_equal(wav, wav2)
    # oracle pipeline with the same draws
    item = infer.preprocess_input(inp, 'phoneme')
    sample = infer.input_to_batch(item)
    sd = {k: v.detach().cpu() for k, v in infer.model.state_dict().items()}
    oin = {'txt_tokens': sample['txt_tokens'].cpu(), 'spk_embed': sample['spk_ids'].cpu(), 'pitch_midi': sample['pitch_midi'].cpu(),
           'midi_dur': sample['midi_dur'].cpu(), 'is_slur': sample['is_slur'].cpu(), 'lang': sample['lang'].cpu(),
           'speechsing': sample['speechsing'].cpu()}
    f = ofs2.fs2_forward(sd, oin)
    B, T = f['mel2ph'].shape
    n = B * 80 * T
    noise = np.stack([synth.philox_normal(99, 0, n)] + [synth.philox_normal(99, i + 1, n) for i in reversed(range(100))])
    r = omg.mel_gen(sd, oin, torch.from_numpy(noise.reshape(101, B, 80, T)), fs2_out=f)
    pe_cpu = {k: v.detach().cpu() for k, v in infer.pe.state_dict().items()}
    f0 = ope.pitch_extractor_forward(pe_cpu, r['mel_out'])['f0_denorm_pred']
    got_f0 = infer.pe(infer._generate(sample, None)['mel_out'])['f0_denorm_pred'].cpu()
    # voiced/unvoiced decisions near 0 may flip with rounding: compare where both agree on voicing
    agree = (f0 > 0) == (got_f0 > 0)
    assert agree.float().mean() > 0.9
    ri = torch.from_numpy(np.random.RandomState(99).uniform(size=(B, 9)).astype(np.float32))
    nz = torch.from_numpy(synth.philox_normal(99, 0x4E5346, B * T * 256 * 9).reshape(B, T * 256, 9))
    want = onsf.nsf_hifigan_forward(nsd, r['mel_out'].transpose(1, 2), got_f0, ri, nz, hcfg)
    assert maxabs(wav, want.reshape(-1)) <= 5e-3
    # the batched entry point drives the same PitchExtractor -> NSF vocoder chain (it used to call the vocoder without f0)
    wb = infer.forward_batch([item], seed=99)
    assert len(wb) == 1 and wb[0].shape == wav.shape and maxabs(wb[0], wav) <= 1e-6
    items = [infer.preprocess_input(_item(n, s), 'phoneme') for n, s in ((8, 5), (5, 6), (10, 7))]
    wb = infer.forward_batch(items, seed=99)
    assert len(wb) == 3 and all(np.isfinite(w).all() and w.ndim == 1 and w.size > 0 for w in wb)
