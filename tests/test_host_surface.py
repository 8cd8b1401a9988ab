"""CPU: the host-side surface the reference exposes for this path — hparams chain semantics, checkpoint discovery,
vocoder registry, token ids, note parsing.  No GPU, no compute calls."""
import os

import pytest
import torch
import yaml

from bisinger_amd import hparams as hp_mod
from bisinger_amd.ckpt import latest_ckpt, load_ckpt
from bisinger_amd.hparams import hparams, set_hparams
from bisinger_amd.text_encoder import TokenTextEncoder


def test_hparams_chain_override_cast_and_saved_config(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    os.makedirs('cfg/sub')
    yaml.safe_dump({'a': 1, 'nested': {'x': 1, 'y': 2}, 'flag': False, 'lst': [1, 2], 'lr': 0.5}, open('cfg/base.yaml', 'w'))
    yaml.safe_dump({'base_config': './base.yaml', 'a': 2, 'nested': {'y': 3}, 'name': 'mid'}, open('cfg/mid.yaml', 'w'))
    yaml.safe_dump({'base_config': ['../mid.yaml'], 'name': 'leaf', 'timesteps': 100}, open('cfg/sub/leaf.yaml', 'w'))
    hp = set_hparams('cfg/sub/leaf.yaml', print_hparams=False)
    # depth-first inheritance, later files override, nested dicts merge (utils/hparams.py:16-21, 48-66)
    assert hp['a'] == 2 and hp['nested'] == {'x': 1, 'y': 3} and hp['name'] == 'leaf' and hp['timesteps'] == 100
    assert hparams['a'] == 2 and hparams['work_dir'] == '' and hparams['infer'] is False
    # typed k=v overrides (:90-96)
    hp = set_hparams('cfg/sub/leaf.yaml', hparams_str='a=7,lr=0.25,flag=True,name=zzz', print_hparams=False)
    assert hp['a'] == 7 and isinstance(hp['a'], int) and hp['lr'] == 0.25 and hp['flag'] is True and hp['name'] == 'zzz'
    # exp_name: work_dir + config saved once, then the saved config wins over the chain (:70-87, :98-101)
    hp = set_hparams('cfg/sub/leaf.yaml', exp_name='exp1', hparams_str='a=9', print_hparams=False)
    assert hp['work_dir'] == 'checkpoints/exp1' and os.path.exists('checkpoints/exp1/config.yaml')
    hp = set_hparams('cfg/sub/leaf.yaml', exp_name='exp1', print_hparams=False)
    assert hp['a'] == 9
    # --reset semantics are only reachable through the CLI flags; the saved file itself is plain YAML of the merged dict
    assert yaml.safe_load(open('checkpoints/exp1/config.yaml'))['a'] == 9
    # global_hparams=False leaves the global dict alone
    before = dict(hparams)
    other = set_hparams('cfg/base.yaml', global_hparams=False, print_hparams=False)
    assert other['a'] == 1 and {k: hparams[k] for k in before} == before


def test_package_configs_resolve():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hp = set_hparams(os.path.join(root, 'bisinger_amd/configs/bisinger_diff100.yaml'), print_hparams=False)
    assert (hp['timesteps'], hp['K_step'], hp['max_beta'], hp['hidden_size'], hp['residual_layers']) == (100, 100, 0.06, 256, 20)
    hp = set_hparams(os.path.join(root, 'bisinger_amd/configs/bisinger_diff.yaml'), print_hparams=False)
    assert (hp['timesteps'], hp['pndm_speedup'], hp['max_beta']) == (1000, 5, 0.02) and len(hp['spec_min']) == 80


def test_load_ckpt_picks_highest_step_and_strips_prefix(tmp_path):
    m = torch.nn.Linear(3, 2)
    for step, scale in ((100, 1.0), (20000, 2.0), (3000, 3.0)):
        sd = {'model.weight': torch.full((2, 3), scale), 'model.bias': torch.zeros(2), 'other.x': torch.zeros(1)}
        torch.save({'state_dict': sd, 'global_step': step}, tmp_path / f'model_ckpt_steps_{step}.ckpt')
    assert latest_ckpt(str(tmp_path)).endswith('model_ckpt_steps_20000.ckpt')
    load_ckpt(m, str(tmp_path), 'model')
    assert float(m.weight[0, 0]) == 2.0
    load_ckpt(m, str(tmp_path / 'model_ckpt_steps_3000.ckpt'), 'model')      # a file path is taken as is
    assert float(m.weight[0, 0]) == 3.0
    with pytest.raises(AssertionError):
        load_ckpt(m, str(tmp_path / 'nowhere'), 'model')
    assert load_ckpt(m, str(tmp_path / 'nowhere'), 'model', force=False) is None
    # strict=False drops shape-mismatched entries
    torch.save({'state_dict': {'model.weight': torch.zeros(5, 5), 'model.bias': torch.ones(2)}}, tmp_path / 'model_ckpt_steps_99999.ckpt')
    load_ckpt(m, str(tmp_path), 'model', strict=False)
    assert float(m.bias[0]) == 1.0 and float(m.weight[0, 0]) == 3.0


def test_vocoder_registry():
    from bisinger_amd import vocoders
    assert vocoders.get_vocoder_cls({'vocoder': 'HifiGAN'}) is vocoders.HifiGAN
    assert vocoders.get_vocoder_cls({'vocoder': 'hifigan'}) is vocoders.HifiGAN
    assert vocoders.get_vocoder_cls({'vocoder': 'vocoders.hifigan.HifiGAN'}) is vocoders.HifiGAN   # the reference's dotted path

    @vocoders.register_vocoder
    class Dummy(vocoders.BaseVocoder):
        pass
    assert vocoders.get_vocoder_cls({'vocoder': 'dummy'}) is Dummy
    assert vocoders.get_vocoder_cls({'vocoder': 'bisinger_amd.vocoders.HifiGAN'}) is vocoders.HifiGAN


def test_token_text_encoder_ids_and_notes():
    enc = TokenTextEncoder(None, vocab_list=['<AP>', '<SP>'] + [f'p{i}' for i in range(60)], replace_oov=',')
    assert len(enc) == 65 and enc.pad() == 0 and enc.eos() == 1 and enc.unk() == 2
    assert enc.encode('<AP> p0 p59') == [3, 5, 64]
    assert enc.decode([3, 5, 0, 64], strip_padding=True) == '<AP> p0 p59'
    from bisinger_amd.infer import note_to_midi
    assert [note_to_midi(n) for n in ('C4', 'C#4', 'Db4', 'B3', 'A0', 'G9')] == [60, 61, 61, 59, 21, 127]
    with pytest.raises(ValueError):
        note_to_midi('H2')


def test_drop_in_state_dicts_match_reference_lists(sd_spec):
    """Every drop-in module reproduces the reference's state_dict names, shapes and order (CPU construction only)."""
    from tests.util import use_config
    hp = use_config()
    from bisinger_amd.candidate_decoder import FFT
    from bisinger_amd.diffnet import DIFF_DECODERS
    from bisinger_amd.diffusion import GaussianDiffusion
    from bisinger_amd.hifigan import HifiGanGenerator
    from bisinger_amd.pe import PitchExtractor

    class Enc:
        def __len__(self):
            return 65

        def pad(self):
            return 0
    lst = lambda m: [[k, list(v.shape)] for k, v in m.state_dict().items()]
    gd = GaussianDiffusion(Enc(), 80, DIFF_DECODERS['wavenet'](hp), timesteps=100, K_step=100, spec_min=hp['spec_min'], spec_max=hp['spec_max'])
    assert lst(gd) == sd_spec['GaussianDiffusion']
    assert lst(FFT(256, 4, 9, 2)) == sd_spec['FFT']
    assert lst(PitchExtractor()) == sd_spec['PitchExtractor']
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, 'bisinger_amd/configs/hifigan.yaml')))
    g = HifiGanGenerator(cfg)
    assert lst(g) == sd_spec['HifiGanGenerator_weight_norm']
    g.remove_weight_norm()
    assert lst(g) == sd_spec['HifiGanGenerator_folded']
    assert lst(HifiGanGenerator(dict(cfg, use_pitch_embed=True))) == sd_spec['HifiGanGenerator_nsf_weight_norm']
    # the product modules refuse to compute on the CPU
    from bisinger_amd import _lib
    with pytest.raises(_lib.BsgError):
        gd.denoise_fn(torch.zeros(1, 1, 80, 8), torch.zeros(1, dtype=torch.long), torch.zeros(1, 256, 8))


def test_diffnet_compute_dtype_option():
    """`diff_compute_dtype` (extension, not in the reference): default fp32, read from hparams, validated on the host
    before any library call; the state_dict is unaffected by it."""
    import pytest
    from bisinger_amd import _lib
    from bisinger_amd.hparams import hparams, set_hparams
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    set_hparams(os.path.join(root, 'bisinger_amd', 'configs', 'bisinger_diff100.yaml'), print_hparams=False)
    from bisinger_amd.diffnet import DiffNet
    a = DiffNet(80)
    assert a.compute_dtype == 'fp32'
    hparams['diff_compute_dtype'] = 'bf16'
    try:
        b = DiffNet(80)
    finally:
        hparams.pop('diff_compute_dtype')
    assert b.compute_dtype == 'bf16'
    assert list(a.state_dict().keys()) == list(b.state_dict().keys())
    a.set_compute('bf16')
    assert a.compute_dtype == 'bf16'
    with pytest.raises(_lib.BsgError):
        a.set_compute('int8')


def test_length_bucketing_matches_reference_batch_by_size():
    """bucket_by_size == the reference's batch_by_size over size-ordered indices (tests/golden/buckets.json, made by
    tools/make_golden_cfg.py from utils/__init__.py:90-143); every index exactly once; budget respected"""
    import json
    from bisinger_amd.infer import bucket_by_size
    cases = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'buckets.json')))
    assert len(cases) >= 5
    for c in cases:
        got = bucket_by_size(c['lengths'], c['max_frames'], c['max_sentences'])
        assert got == c['batches']
        assert sorted(i for b in got for i in b) == list(range(len(c['lengths'])))
        for b in got:
            if c['max_frames'] is not None:
                assert len(b) * max(c['lengths'][i] for i in b) <= c['max_frames']
            if c['max_sentences'] is not None:
                assert len(b) <= c['max_sentences']
    import pytest
    with pytest.raises(ValueError):
        bucket_by_size([10, 5000], max_frames=4000)
    assert bucket_by_size([], 100) == []
    assert bucket_by_size([3, 9, 4]) == [[1, 2, 0]]
