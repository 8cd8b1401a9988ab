"""GPU parity for the surface gaps closed in round 4, against outputs of the reference (tests/golden/r4.npz,
tools/make_golden_r4.py) and the oracle: ResBlock2 generators, the original release's vocoder checkpoint layout, PLMS over a generic
denoiser."""
import json
from collections import OrderedDict

import numpy as np
import pytest
import torch

from bisinger_amd import synth
from bisinger_amd.hparams import hparams
from oracle import candidate_decoder as ocd, diffusion as odf, hifigan as ohg
from tests.util import cpu_sd, maxabs, use_config

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
T_ = torch.from_numpy


def test_hifigan_resblock2(gold, sd_spec):
    """h['resblock'] = '2' (TB/modules/hifigan/hifigan.py:70-91, selected at :117): one dilated conv per dilation with its own residual;
    kernels 3 / 5 / 7, dilations up to 12, three upsampling stages — the reference's output, then a longer batch against the oracle."""
    from bisinger_amd.hifigan import HifiGanGenerator
    cfg = sd_spec['hifigan_rb2_cfg']
    g = HifiGanGenerator(cfg)
    assert [[k, list(v.shape)] for k, v in g.state_dict().items()] == sd_spec['HifiGanGenerator_rb2_weight_norm']
    spec = OrderedDict((k, tuple(s)) for k, s in sd_spec['HifiGanGenerator_rb2_weight_norm'])
    sd = {k: T_(v) for k, v in synth.synth_state_dict(spec, 27).items()}
    g.load_state_dict(sd, strict=True)
    g = g.cuda()
    rs = np.random.RandomState(31)
    mel = (rs.standard_normal((2, 80, 37)) * 1.5 - 3.0).astype(np.float32)
    y = g(T_(mel).cuda())
    assert y.shape == (2, 1, 37 * 256)
    assert maxabs(y, gold('r4')['rb2.wav']) <= 2e-5
    g.remove_weight_norm()                                    # folded layout: same result
    assert maxabs(g(T_(mel).cuda()), gold('r4')['rb2.wav']) <= 2e-5
    mel = (np.random.RandomState(5).standard_normal((3, 80, 150)) * 1.5 - 3.0).astype(np.float32)
    want = ohg.hifigan_forward(sd, T_(mel), cfg)
    assert maxabs(g(T_(mel).cuda()), want) <= 5e-5


def test_vocoder_release_checkpoint_layout(tmp_path, gold, sd_spec):
    """<vocoder_ckpt>/config.json + generator_v1 (ckpt['generator']): the layout of the original HiFi-GAN release, taken when there is no
    config.yaml (TB/vocoders/hifigan.py:21-24,47-51).  Same formula weights as the fixture; spec2wav's contract mel [T, 80] -> wav [T hop]."""
    from bisinger_amd.vocoders import HifiGAN, get_vocoder_cls
    cfg = sd_spec['hifigan_v1_json']
    spec = OrderedDict((k, tuple(s)) for k, s in sd_spec['HifiGanGenerator_weight_norm'])
    json.dump(cfg, open(tmp_path / 'config.json', 'w'))
    torch.save({'generator': {k: T_(v) for k, v in synth.synth_state_dict(spec, 7).items()}}, tmp_path / 'generator_v1')
    use_config()
    hparams['vocoder_ckpt'] = str(tmp_path)
    hparams['vocoder'] = 'vocoders.hifigan.HifiGAN'
    assert get_vocoder_cls(hparams) is HifiGAN
    voc = HifiGAN()
    rs = np.random.RandomState(33)
    mel = (rs.standard_normal((1, 80, 16)) * 1.5 - 3.0).astype(np.float32)
    wav = voc.spec2wav(mel[0].T)
    assert wav.shape == (16 * 256,)
    assert maxabs(wav, gold('r4')['v1json.wav'].reshape(-1)) <= 2e-5


class _Enc:
    def __len__(self):
        return 65

    def pad(self):
        return 0


def test_plms_over_a_generic_denoiser(gold, sd_spec):
    """pndm_speedup with DIFF_DECODERS['fft'] (TB/usr/diff/shallow_diffusion_tts.py:168-201,258-264 works with any denoise_fn): the
    reference's B = 1 loop (20 iterations, 21 evaluations of the FFT denoiser), then B = 2 against the oracle's batched semantics."""
    use_config('diff_decoder_type=fft,pndm_speedup=5')
    try:
        from bisinger_amd.diffnet import DIFF_DECODERS
        from bisinger_amd.diffusion import GaussianDiffusion
        net = DIFF_DECODERS[hparams['diff_decoder_type']](hparams)
        spec = OrderedDict((k, tuple(s)) for k, s in sd_spec['FFT'])
        net.load_state_dict({k: T_(v) for k, v in synth.synth_state_dict(spec, seed=17).items()}, strict=False)
        model = GaussianDiffusion(_Enc(), 80, net, timesteps=100, K_step=100, spec_min=hparams['spec_min'], spec_max=hparams['spec_max']).cuda()
        rs = np.random.RandomState(43)
        cond = rs.standard_normal((1, 256, 40)).astype(np.float32)
        xT = rs.standard_normal((1, 1, 80, 40)).astype(np.float32)
        got = model.sample(T_(cond).cuda(), T_(xT).cuda().contiguous())
        assert maxabs(got, gold('r4')['plmsfft.x0']) <= 5e-4
        sd = cpu_sd(model.denoise_fn)
        rs = np.random.RandomState(44)
        cond2 = T_(rs.standard_normal((2, 256, 56)).astype(np.float32))
        x2 = T_(rs.standard_normal((2, 1, 80, 56)).astype(np.float32))
        want = odf.plms_sample(odf.make_schedule(100, 'linear', 0.06), lambda x_, t_: ocd.fft_denoiser_forward(sd, x_, t_, cond2), x2, 100, 5)
        got2 = model.sample(cond2.cuda(), x2.cuda().contiguous())
        assert maxabs(got2, want) <= 5e-4
    finally:
        use_config()
