"""GPU: edge shapes and error paths of the C ABI (tiny / ragged / long inputs; misuse raises instead of faulting)."""
import numpy as np
import pytest
import torch

from bisinger_amd import _lib, synth
from bisinger_amd.hparams import hparams
from oracle import diffnet as odn, fs2 as ofs2, melgen as omg
from tests.util import cpu_sd, load_formula_weights, maxabs, use_config

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
T_ = torch.from_numpy


class _Enc:
    def __len__(self):
        return 65

    def pad(self):
        return 0


@pytest.fixture(scope='module')
def model():
    use_config()
    from bisinger_amd.diffnet import DIFF_DECODERS
    from bisinger_amd.diffusion import GaussianDiffusion
    m = GaussianDiffusion(_Enc(), 80, DIFF_DECODERS['wavenet'](hparams), timesteps=100, K_step=100,
                          spec_min=hparams['spec_min'], spec_max=hparams['spec_max'])
    load_formula_weights(m, 0, synth.DIFFNET_GAIN)
    return m.cuda()


@pytest.mark.parametrize('B,T', [(1, 1), (2, 5), (1, 33), (3, 17)])
def test_diffnet_tiny_T(model, B, T):
    """Frames fewer than the dilation halo / one partial tile."""
    sd = cpu_sd(model)
    rs = np.random.RandomState(B * 100 + T)
    x = rs.standard_normal((B, 1, 80, T)).astype(np.float32)
    cond = rs.standard_normal((B, 256, T)).astype(np.float32)
    t = rs.randint(0, 100, size=(B,)).astype(np.int64)
    got = model.denoise_fn(T_(x).cuda(), T_(t).cuda(), T_(cond).cuda())
    want = odn.diffnet_forward(sd, T_(x), T_(t), T_(cond), 'denoise_fn.')
    assert maxabs(got, want) <= 5e-5


@pytest.mark.parametrize('B,Tt,T', [(1, 1, 1), (1, 2, 3), (2, 3, 40)])
def test_melgen_tiny(model, B, Tt, T):
    inp = synth.synth_inputs(B, Tt, T, seed=3)
    noise = synth.synth_noise(100, B, 80, T, seed=4)
    d = {k: T_(v).cuda() for k, v in inp.items()}
    kw = {k: d[k] for k in ('pitch_midi', 'midi_dur', 'is_slur', 'lang', 'speechsing')}
    got = model(d['txt_tokens'], mel2ph=d['mel2ph'], spk_embed=d['spk_embed'], infer=True, noise=T_(noise), **kw)
    want = omg.mel_gen(cpu_sd(model), {k: T_(v) for k, v in inp.items()}, T_(noise))
    assert maxabs(got['mel_out'], want['mel_out']) <= 1e-3
    assert maxabs(got['fs2_mel'], want['fs2_mel']) <= 1e-4


def test_fs2_long_ragged(model):
    """T = 2500 frames (several attention tiles, odd lengths, ~half of max_frames), ragged batch."""
    inp = synth.synth_inputs(2, 250, 2500, seed=7, ragged=True)
    d = {k: T_(v).cuda() for k, v in inp.items()}
    kw = {k: d[k] for k in ('pitch_midi', 'midi_dur', 'is_slur', 'lang', 'speechsing')}
    got = model.fs2(d['txt_tokens'], d['mel2ph'], d['spk_embed'], skip_decoder=False, infer=True, **kw)
    want = ofs2.fs2_forward(cpu_sd(model), {k: T_(v) for k, v in inp.items()})
    assert maxabs(got['decoder_inp'], want['decoder_inp']) <= 1e-4
    assert maxabs(got['mel_out'], want['mel_out']) <= 3e-4


def test_misuse_raises_instead_of_faulting(model):
    net = model.denoise_fn
    lib = _lib.load()
    cond = torch.randn(2, 256, 40, device='cuda')
    net.prepare(cond)
    x = torch.randn(3, 80, 40, device='cuda')          # B does not match the bound condition
    t = torch.zeros(3, dtype=torch.long, device='cuda')
    eps = torch.empty_like(x)
    rc = lib.bsg_diffnet_forward(net._h, _lib.ptr(x), _lib.ptr(t), _lib.ptr(eps), 3, 40, _lib.stream_ptr())
    assert rc != 0 and b'bsg_diffnet_prepare' in lib.bsg_last_error()
    with pytest.raises(_lib.BsgError):
        _lib.check(lib.bsg_diffnet_forward(net._h, None, _lib.ptr(t), _lib.ptr(eps), 2, 40, _lib.stream_ptr()), 'fwd')
    # timestep outside the schedule
    model.K_step = 101
    try:
        with pytest.raises(_lib.BsgError):
            model.sample(cond, torch.randn(2, 1, 80, 40, device='cuda'))
    finally:
        model.K_step = 100
    # CPU tensors are refused (no CPU path)
    from bisinger_amd.diffnet import DiffNet
    with pytest.raises(_lib.BsgError):
        DiffNet(80)(torch.randn(1, 1, 80, 8), torch.zeros(1, dtype=torch.long), torch.randn(1, 256, 8))
    # empty utterance from the duration predictor is reported, not launched
    with pytest.raises((_lib.BsgError, AssertionError, RuntimeError)):
        lib2 = model.fs2
        enc = dict(dur=torch.zeros(1, 4, dtype=torch.long, device='cuda'), txt=torch.ones(1, 4, dtype=torch.long, device='cuda'))
        lib2.regulate(enc)


def test_reload_weights_rebuilds_handle(model):
    """load_state_dict after first use must be picked up (the handle caches packed weights)."""
    net = model.denoise_fn
    x = torch.randn(1, 1, 80, 16, device='cuda')
    cond = torch.randn(1, 256, 16, device='cuda')
    t = torch.tensor([5], device='cuda')
    a = net(x, t, cond).clone()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    sd2 = {k: (v * 1.5 if k == 'output_projection.weight' else v) for k, v in sd.items()}
    net.load_state_dict(sd2)
    b = net(x, t, cond).clone()
    net.load_state_dict(sd)
    c = net(x, t, cond)
    assert maxabs(a, b) > 1e-3 and maxabs(a, c) == 0.0
