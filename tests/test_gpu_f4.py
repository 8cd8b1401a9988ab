"""GPU parity, SURVEY.md §8 row f4: the FFT candidate denoiser against the golden produced by the reference."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from bisinger_amd import synth
from bisinger_amd.hparams import hparams
from oracle import candidate_decoder as ocd, diffusion as odf
from tests.util import cpu_sd, maxabs, use_config

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
T_ = torch.from_numpy


class _Enc:
    def __len__(self):
        return 65

    def pad(self):
        return 0


def test_fft_denoiser_golden_and_sampler(gold, sd_spec):
    use_config('diff_decoder_type=fft')
    from bisinger_amd.diffnet import DIFF_DECODERS
    from bisinger_amd.diffusion import GaussianDiffusion
    net = DIFF_DECODERS[hparams['diff_decoder_type']](hparams)
    assert [[k, list(v.shape)] for k, v in net.state_dict().items()] == sd_spec['FFT']
    spec = OrderedDict((k, tuple(s)) for k, s in sd_spec['FFT'])
    net.load_state_dict({k: T_(v) for k, v in synth.synth_state_dict(spec, seed=17).items()}, strict=False)
    model = GaussianDiffusion(_Enc(), 80, net, timesteps=100, K_step=100, spec_min=hparams['spec_min'], spec_max=hparams['spec_max']).cuda()
    g = gold('f4')
    rs = np.random.RandomState(41)
    for tag, (B, T) in {'B2T40': (2, 40), 'B1T77': (1, 77)}.items():
        x = rs.standard_normal((B, 1, 80, T)).astype(np.float32)
        cond = rs.standard_normal((B, 256, T)).astype(np.float32)
        t = rs.randint(0, 100, size=(B,)).astype(np.int64)
        eps = model.denoise_fn(T_(x).cuda(), T_(t).cuda(), T_(cond).cuda())
        assert maxabs(eps, g[f'{tag}.eps']) <= 2e-4, tag
    # a short DDPM run with the generic step (bsg_ddpm_step) vs the oracle
    sd = cpu_sd(model.denoise_fn)
    noise = synth.synth_noise(6, 1, 80, 77, seed=8)
    cond_t = T_(cond)
    den = lambda x_, t_: ocd.fft_denoiser_forward(sd, x_, t_, cond_t)
    sch = odf.make_schedule(100, 'linear', 0.06)
    want = T_(noise[0][:, None])
    for k in range(6):
        want = odf.p_sample(sch, den, want, torch.full((1,), 99 - k, dtype=torch.long), T_(noise[1 + k][:, None]))
    got = model.sample(cond_t.cuda(), T_(noise[0][:, None]).cuda().contiguous(), noise=T_(noise[1:]).cuda(), n_steps=6)
    assert maxabs(got, want) <= 5e-4
    use_config()
