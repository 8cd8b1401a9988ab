"""GPU: round 6 — the range guard is per handle, has a middle tier, and the default launches hold on He-gain and heavy-tailed weights.


VERDICT r05 item 4: (a) one model tripping its guard leaves every other handle of the process alone; (b) a range event of the 16-row stack
launch (|x| >= 3750) is retried on the 32-row launch (|x + d| < 60000), not on the fp32 matrix pipe; (d) weights as the reference's
constructors draw them (kaiming_normal_, usr/diff/net.py:47-50: std = sqrt(2 / fan_in)) and Student-t(3) weights, 100 sampler steps against
the oracle, without a single range event."""
import math
import warnings

import numpy as np
import pytest
import torch

from bisinger_amd import _lib, synth
from bisinger_amd.hparams import hparams
from oracle import diffnet as odn, diffusion as odf
from tests.util import cpu_sd, load_formula_weights, maxabs, use_config

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
T_ = torch.from_numpy


class _Enc:
    def __len__(self):
        return 65

    def pad(self):
        return 0


def _diffusion(seed=0):
    use_config()
    from bisinger_amd.diffnet import DIFF_DECODERS
    from bisinger_amd.diffusion import GaussianDiffusion
    m = GaussianDiffusion(_Enc(), 80, DIFF_DECODERS['wavenet'](hparams), timesteps=100, K_step=100,
                          spec_min=hparams['spec_min'], spec_max=hparams['spec_max'])
    load_formula_weights(m, seed, synth.DIFFNET_GAIN)
    return m.cuda()


def _inputs(B, Tt, T, seed):
    inp = synth.synth_inputs(B, Tt, T, seed=seed)
    d = {k: T_(v).cuda() for k, v in inp.items()}
    kw = {k: d[k] for k in ('pitch_midi', 'midi_dur', 'is_slur', 'lang', 'speechsing')}
    return d, kw


def test_one_models_range_event_leaves_the_other_models_handles_alone():
    """Two GaussianDiffusion models (each: an FS2 handle and a DiffNet handle) and a vocoder in one process.  Model A's token embedding is
    scaled so that its FS2 GEMM operands leave the fp16 range of the split (|v| >= 4062): A's call warns, repeats with A's FS2 handle on
    the fp32 matrix pipe and returns finite mels.  Model B and the vocoder never notice: no strike, their GEMM switch on, the same launch
    form and bit-identical results before and after."""
    from bisinger_amd.hifigan import HifiGanGenerator
    A, Bm = _diffusion(0), _diffusion(1)
    d, kw = _inputs(2, 12, 64, 4)
    noise = T_(synth.synth_noise(100, 2, 80, 64, seed=5)).cuda()
    call = lambda m: m(d['txt_tokens'], mel2ph=d['mel2ph'], spk_embed=d['spk_embed'], ref_mels=None, infer=True, noise=noise, **kw)['mel_out'].clone()
    ref_b = call(Bm)
    path_b = Bm.denoise_fn.last_path()
    with torch.no_grad():
        A.fs2.encoder_embed_tokens.weight.mul_(3e4)      # sqrt(H) * 3e4 * N(0, 1/16): operands of the ESM's query projection ~ 1e5
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        out_a = call(A)
    msgs = [str(m.message) for m in w]
    assert any('FastSpeech2MIDI' in m and 'fp16 range of the split-fp16 GEMMs' in m for m in msgs), msgs
    assert torch.isfinite(out_a).all()
    # the event was FS2's; A's OWN denoiser may count one too in the same pass (a front that left the range hands it a non-finite condition:
    # cause and effect cannot be told apart inside one pass, so both of A's handles repeat on the fp32 matrix pipe) — never B's
    assert A.fs2.gemm_range_strikes == 1 and A.denoise_fn.gemm_range_strikes <= 1
    assert A.fs2.gemm_split_enabled()                                                   # back on the split form for the next call (strike 1 of 3)
    # the other model: untouched
    assert Bm.fs2.gemm_range_strikes == 0 and Bm.denoise_fn.gemm_range_strikes == 0
    assert Bm.fs2.gemm_split_enabled() and Bm.denoise_fn.gemm_split_enabled()
    with warnings.catch_warnings(record=True) as w2:
        warnings.simplefilter('always')
        again_b = call(Bm)
    assert not w2, [str(m.message) for m in w2]
    assert Bm.denoise_fn.last_path() == path_b and torch.equal(again_b, ref_b)
    # three strikes keep A's FS2 handle on the fp32 matrix pipe; B's stays on the split form
    for _ in range(2):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            call(A)
    assert A.fs2.gemm_range_strikes == 3 and not A.fs2.gemm_split_enabled()
    assert Bm.fs2.gemm_split_enabled() and _lib.load().bsg_abi_version() >= 7
    with warnings.catch_warnings(record=True) as w3:
        warnings.simplefilter('always')
        out_a2 = call(A)                                  # no event any more: the handle's GEMMs are fp32 products
    assert not w3 and maxabs(out_a2, out_a) <= 1e-4


def test_q_launch_range_event_retries_on_the_32_row_launch():
    """|x| >= 3750 inside the 16-row stack launch (its conv image holds 16 x as fp16 planes): the handle takes the 32-row launch — still the
    16-bit matrix pipe, limit |x + d| < 60000 — for the bound condition, not the fp32 pipe; the result equals the fp32-pipe reference."""
    use_config()
    from bisinger_amd.diffnet import DiffNet
    net = load_formula_weights(DiffNet(80), 0, synth.DIFFNET_GAIN, prefix='denoise_fn.').cuda()
    ref = load_formula_weights(DiffNet(80), 0, synth.DIFFNET_GAIN, prefix='denoise_fn.').cuda()
    B, T = 10, 900                                          # a whole-tile stack launch (no part form at this size)
    rs = np.random.RandomState(2)
    # the GEMM operands stay inside THEIR range (|x| < 4062 for the input projection) while the residual stream leaves the 16-row launch's:
    # W_in x 10 on both nets, x ~ 400 N(0, 1): x_0 = relu(W_in x) ~ 4000 N(0, 1), max ~ 1.6e4 — beyond 3750, inside 60000
    with torch.no_grad():
        net.input_projection.weight.mul_(10.0)
        ref.input_projection.weight.mul_(10.0)
    x = T_((rs.standard_normal((B, 1, 80, T)) * 400).astype(np.float32)).cuda()
    t = torch.full((B,), 37, device='cuda', dtype=torch.long)
    cond = T_(rs.standard_normal((B, 256, T)).astype(np.float32)).cuda()
    small = net(x * 1e-4, t, cond)
    assert net.last_path().startswith('stack_h2q')
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        y = net(x, t, cond).clone()
    msgs = [str(m.message) for m in w]
    assert any('|x| >= 3750' in m and '32-row launch' in m for m in msgs), msgs
    assert net.last_path() == 'stack_h2', net.last_path()                 # the middle tier held: no fp32-pipe warning, no third pass
    assert not any('fp32 matrix pipe' in m for m in msgs), msgs
    assert net.gemm_range_strikes == 0 and net.gemm_split_enabled()
    ref.set_split_fp16(False)
    yr = ref(x, t, cond)
    assert not ref.last_path().startswith('stack_h2')
    # (activations of 1e4 carry an fp32 ulp of 1e-3 through the residual stream: two fp32-grade forms agree to ~1e-4 on eps here)
    assert maxabs(y, yr) <= 2e-4 * max(1.0, float(yr.abs().max()))
    # another condition: the 16-row launch comes back
    cond2 = cond * 0.5
    net(x * 1e-4, t, cond2)
    assert net.last_path().startswith('stack_h2q')
    del small


def _reference_init(net, kind, seed):
    """'he': every Conv1d as the reference's constructors draw it (kaiming_normal_: N(0, 2 / fan_in), usr/diff/net.py:47-50), Linear layers
    as torch's default (U(+-1/sqrt(fan_in))); the final output projection — zeros in the reference (:104) — He as well, or eps would be 0.
    't3': the same scales with Student-t(3) draws (heavy tails: single weights 10 x the layer's std)."""
    rs = np.random.RandomState(seed)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if p.dim() >= 2:
                fan_in = int(np.prod(p.shape[1:]))
                std = math.sqrt(2.0 / fan_in)
                if kind == 't3':
                    v = rs.standard_t(3, size=tuple(p.shape)) * (std / math.sqrt(3.0))
                elif 'conv' in name or 'projection' in name and p.dim() == 3:
                    v = rs.standard_normal(tuple(p.shape)) * std
                else:
                    v = rs.uniform(-1, 1, size=tuple(p.shape)) / math.sqrt(fan_in)
                p.copy_(T_(v.astype(np.float32)))
            else:
                p.copy_(T_((rs.uniform(-1, 1, size=tuple(p.shape)) * 0.05).astype(np.float32)))


@pytest.mark.parametrize('kind', ['he', 't3'])
def test_default_launches_hold_on_reference_init_weights(kind):
    """100 DDPM steps at B = 2, T = 256 with He-gain weights (what the reference's own constructors draw) and Student-t(3) weights, on the part
    form this shape takes by default and on the 16-row stack launch of the headline (part forms off), with the output projection (zeros in
    the reference, net.py:104) at He gain and at 0.1 of it: no range event, no strike, the launch stays a split-fp16 one; one evaluation
    within 2e-6 (relative to max |eps|) of the oracle and the 100-step trajectory far inside north_star's 1e-3 (measured: 1e-6 .. 7e-6)."""
    B, T = 2, 256
    rs = np.random.RandomState(3)
    cond = T_(rs.standard_normal((B, 256, T)).astype(np.float32))
    noise = T_(synth.synth_noise(100, B, 80, T, seed=9))
    t = torch.full((B,), 63, dtype=torch.long)
    sch = odf.make_schedule(100, 'linear', 0.06)
    devs = {}
    for out_gain in (1.0, 0.1):
        m = _diffusion(0)
        net = m.denoise_fn
        _reference_init(net, kind, 7)
        with torch.no_grad():
            net.output_projection.weight.mul_(out_gain)
        sd = cpu_sd(net)
        want_e = odn.diffnet_forward(sd, noise[0][:, None], t, cond)
        want = odf.ddpm_sample(sch, lambda xx, tt: odn.diffnet_forward(sd, xx, tt, cond), noise[0][:, None], noise[1:][:, :, None], 100)
        for parts in (True, False):
            _lib.check(_lib.load().bsg_diffnet_set_parts(net.handle(), int(parts)), 'bsg_diffnet_set_parts')
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter('always')
                e = net(noise[0][:, None].cuda(), t.cuda(), cond.cuda())
                path1 = net.last_path()
                x = m.sample(cond.cuda(), noise[0][:, None].cuda().contiguous(), noise=noise[1:].cuda(), n_steps=100)
                path = net.last_path()
            assert not w, [str(v.message) for v in w]
            assert path1.startswith('stack_h2') and path.startswith('stack_h2'), (path1, path)
            assert parts or (path.startswith('stack_h2q') and path1.startswith('stack_h2q')), (path1, path)
            give, rng = net.take_health()
            assert give == 0 and rng == 0 and net.gemm_range_take() == 0 and net.gemm_range_strikes == 0
            assert torch.isfinite(x).all()
            d1 = maxabs(e, want_e)
            assert d1 <= 2e-6 * max(1.0, float(want_e.abs().max())), (d1, float(want_e.abs().max()))
            dev = maxabs(x, want)
            print(f'{kind} output-projection gain {out_gain} parts={parts}: paths {path1} / {path}; one evaluation vs oracle {d1:.2e} (max |eps| '
                  f'{float(want_e.abs().max()):.2f}); 100 steps vs oracle {dev:.2e}')
            devs[(out_gain, parts)] = dev
    print(devs)
    assert max(devs.values()) <= 1e-4      # (north_star: 1e-3)


def test_every_handle_kind_guards_itself_and_only_itself(hifigan_sd, sd_spec):
    """The per-handle range guard on the three other handle kinds — HiFi-GAN generator, PitchExtractor, FFT candidate denoiser — beside a
    DiffNet that must not notice: each gets an input that leaves the fp16 range of its split-fp16 products (|v| >= 4062), warns naming
    ITSELF, returns what the same model gives with its products on the fp32 matrix pipe from the start, counts one strike, and is back on
    the split form afterwards; the bystander's strike count, switch and launch form never change."""
    import yaml
    from collections import OrderedDict
    from bisinger_amd.candidate_decoder import FFT
    from bisinger_amd.diffnet import DiffNet
    from bisinger_amd.hifigan import HifiGanGenerator
    from bisinger_amd.pe import PitchExtractor
    from tests.util import ROOT
    hp = use_config()
    hp.update(pitch_type='frame', use_uv=True, pitch_norm='log')
    rs = np.random.RandomState(12)
    by = load_formula_weights(DiffNet(80), 0, synth.DIFFNET_GAIN, prefix='denoise_fn.').cuda()
    bx = T_(rs.standard_normal((2, 1, 80, 96)).astype(np.float32)).cuda()
    bt = torch.full((2,), 11, device='cuda', dtype=torch.long)
    bc = T_(rs.standard_normal((2, 256, 96)).astype(np.float32)).cuda()
    ref_by = by(bx, bt, bc).clone()
    path_by = by.last_path()

    def bystander_untouched():
        assert by.gemm_range_strikes == 0 and by.gemm_split_enabled()
        assert torch.equal(by(bx, bt, bc), ref_by) and by.last_path() == path_by

    def voc():
        g = HifiGanGenerator(yaml.safe_load(open(f'{ROOT}/bisinger_amd/configs/hifigan.yaml')))
        g.load_state_dict(hifigan_sd, strict=True)
        return g.cuda()

    def pe():
        m = PitchExtractor()
        spec = OrderedDict((k, tuple(s)) for k, s in sd_spec['PitchExtractor'])
        w = synth.synth_state_dict(spec, seed=11)
        for k in spec:
            if k.endswith('running_var'):
                w[k] = (0.5 + np.abs(w[k]) * 5).astype(np.float32)
        m.load_state_dict({k: T_(v) for k, v in w.items()}, strict=False)
        return m.cuda()

    def fft():
        m = FFT(256, 4, 9, 2)
        spec = OrderedDict((k, tuple(s)) for k, s in sd_spec['FFT'])
        m.load_state_dict({k: T_(v) for k, v in synth.synth_state_dict(spec, seed=17).items()}, strict=False)
        return m.cuda()

    mel_v = T_((rs.standard_normal((1, 80, 40)) * 1.5 - 3.0).astype(np.float32)).cuda() * 4e3      # conv_pre's operand: |v| ~ 2e4
    mel_p = T_((rs.standard_normal((2, 50, 80)) * 1.5 - 3.0).astype(np.float32)).cuda() * 4e3
    fx = T_(rs.standard_normal((2, 1, 80, 40)).astype(np.float32)).cuda()
    ftt = torch.full((2,), 7, device='cuda', dtype=torch.long)
    fc = T_(rs.standard_normal((2, 256, 40)).astype(np.float32)).cuda() * 2e4                        # the hoisted condition part's operand
    cases = [('HifiGanGenerator', voc, lambda m: m(mel_v)), ('PitchExtractor', pe, lambda m: m(mel_p)['pitch_pred']),
             ('FFT', fft, lambda m: m(fx, ftt, fc))]
    for name, make, call in cases:
        m = make()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            got = call(m).clone()
        msgs = [str(v.message) for v in w]
        assert any(name in s and 'fp16 range of the split-fp16 GEMMs' in s for s in msgs), (name, msgs)
        assert m.gemm_range_strikes == 1 and m.gemm_split_enabled(), name
        ref = make()
        ref.set_gemm_split(False)                       # the fp32 matrix pipe from the first call on
        with warnings.catch_warnings(record=True) as w2:
            warnings.simplefilter('always')
            want = call(ref)
        assert not w2 and ref.gemm_range_strikes == 0, (name, [str(v.message) for v in w2])
        assert torch.isfinite(got).all() and maxabs(got, want) <= 1e-5 * max(1.0, float(want.abs().max())), (name, maxabs(got, want))
        bystander_untouched()


def test_two_host_threads_on_two_models_do_not_share_guard_state():
    """include/bisinger_hip.h: 'distinct handles are independent'; the guard state a compute entry sees is thread-local (bsg::GuardScope in the
    library, _lib._tls in the wrappers).  Two host threads, each with its own model on its own stream, run concurrently (ctypes releases the
    GIL inside the library): thread A's input leaves the fp16 range of its FS2 GEMMs in every call — warned, repeated on the fp32 matrix pipe,
    three strikes — while thread B's calls stay bit-identical to its reference, strike-free and on the split-fp16 GEMMs throughout."""
    import threading
    A, Bm = _diffusion(0), _diffusion(1)
    d, kw = _inputs(2, 12, 64, 4)
    noise = T_(synth.synth_noise(100, 2, 80, 64, seed=5)).cuda()
    call = lambda m: m(d['txt_tokens'], mel2ph=d['mel2ph'], spk_embed=d['spk_embed'], ref_mels=None, infer=True, noise=noise, **kw)['mel_out'].clone()
    ref_b = call(Bm)
    with torch.no_grad():
        A.fs2.encoder_embed_tokens.weight.mul_(3e4)
    torch.cuda.synchronize()
    res, errs = {'a': [], 'b': []}, []

    def worker(tag, m, n):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s), warnings.catch_warnings():
                warnings.simplefilter('ignore')
                for _ in range(n):
                    res[tag].append(call(m))
                s.synchronize()
        except Exception as e:      # noqa: BLE001
            errs.append((tag, repr(e)))

    ta, tb = threading.Thread(target=worker, args=('a', A, 4)), threading.Thread(target=worker, args=('b', Bm, 8))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert not errs, errs
    assert len(res['a']) == 4 and all(bool(torch.isfinite(o).all()) for o in res['a'])
    assert A.fs2.gemm_range_strikes == 3 and not A.fs2.gemm_split_enabled()
    assert all(torch.equal(o, ref_b) for o in res['b'])
    assert Bm.fs2.gemm_range_strikes == 0 and Bm.denoise_fn.gemm_range_strikes == 0 and Bm.fs2.gemm_split_enabled() and Bm.denoise_fn.gemm_split_enabled()
