"""GPU: the bf16-operand configuration of the fused residual layers (BASELINE configs[2]; bsg_diffnet_set_compute).

Not a parity configuration: the <= 1e-3 mel bar belongs to fp32 (test_gpu_*.py).  Two checks pin what it computes:
  (1) against the oracle with the SAME rounding emulated (oracle.diffnet.residual_block(operand_bf16=True)):
      weights, x + d, the hoisted conditioner term, the gated z and the stored skip sum rounded to bf16 (RNE), fp32 accumulation — the kernel must agree up to fp32
      summation order and the rare 1-ulp bf16 flips that a 1e-7 difference before a rounding causes;
  (2) its deviation from the fp32 configuration over a full 100-step sampler run is bounded and printed.
"""
import numpy as np
import pytest
import torch

from bisinger_amd import synth
from oracle import diffnet as odn
from tests.util import cpu_sd, load_formula_weights, maxabs, use_config

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
T_ = torch.from_numpy


@pytest.fixture(scope='module')
def net():
    use_config()
    from bisinger_amd.diffnet import DiffNet
    m = DiffNet(80)
    load_formula_weights(m, 0, synth.DIFFNET_GAIN, prefix='denoise_fn.')
    m = m.cuda()
    yield m
    m.set_compute('fp32')


@pytest.mark.parametrize('B,T', [(2, 64), (3, 77), (1, 31), (5, 333), (2, 1000)])
@pytest.mark.parametrize('layer', [0, 1, 2, 3, 19])
def test_residual_layer_bf16(net, B, T, layer):
    """every dilation, partial last 64-frame tile, T < one tile, first-layer store / last-layer scaling of the skip sum"""
    sd = cpu_sd(net, 'denoise_fn.')
    rs = np.random.RandomState(100 * B + T + layer)
    x = rs.standard_normal((B, 256, T)).astype(np.float32)
    cond = rs.standard_normal((B, 256, T)).astype(np.float32)
    skip0 = rs.standard_normal((B, 256, T)).astype(np.float32)
    t = rs.randint(0, 100, size=(B,)).astype(np.int64)
    d = odn.step_embedding(sd, T_(t), 256, 'denoise_fn.')
    p = f'denoise_fn.residual_layers.{layer}.'
    rx, rskip = odn.residual_block(sd, p, T_(x), T_(cond), d, 2 ** (layer % 4), operand_bf16=True)
    fx, fskip = odn.residual_block(sd, p, T_(x), T_(cond), d, 2 ** (layer % 4))
    scale = 1 / 20 ** 0.5 if layer == 19 else 1.0
    bf = lambda v: v.to(torch.bfloat16).float()
    want_skip = bf((rskip if layer == 0 else bf(T_(skip0)) + rskip) * scale)     # the skip sum is stored as bf16
    net.set_compute('bf16')
    net.prepare(T_(cond).cuda())
    skip = T_(skip0).cuda()
    if layer == 0:
        skip.fill_(float('nan'))
    out = net.residual_layer(layer, T_(x).cuda(), T_(t).cuda(), skip)
    torch.cuda.synchronize()
    net.set_compute('fp32')
    e_x, e_s = maxabs(out, rx), maxabs(skip, want_skip)
    q_x, q_s = maxabs(rx, fx), maxabs(rskip, fskip)          # what the operand rounding itself costs
    print(f'bf16 layer {layer} B={B} T={T}: kernel vs bf16-emulating oracle {e_x:.2e}/{e_s:.2e}; '
          f'rounding cost vs fp32 oracle {q_x:.2e}/{q_s:.2e}')
    assert e_x <= 5e-3 and e_s <= 6.3e-2      # skip: a 1-ulp bf16 flip of a stored value of magnitude < 16
    assert e_x < 0.5 * q_x + 1e-4 or e_x <= 2e-4      # far closer to the emulation than the emulation is to fp32


def test_bf16_does_not_change_fp32_results(net):
    """switching the mode back restores the bit-identical fp32 path"""
    rs = np.random.RandomState(5)
    x = T_(rs.standard_normal((2, 1, 80, 96)).astype(np.float32)).cuda()
    cond = T_(rs.standard_normal((2, 256, 96)).astype(np.float32)).cuda()
    t = torch.tensor([3, 77], device='cuda')
    a = net(x, t, cond).clone()
    net.set_compute('bf16')
    b = net(x, t, cond).clone()
    net.set_compute('fp32')
    c = net(x, t, cond).clone()
    torch.cuda.synchronize()
    assert torch.equal(a, c)
    assert not torch.equal(a, b)
    dev = maxabs(a, b)
    print(f'DiffNet eps, bf16 operands vs fp32: max abs {dev:.3e} (eps rms {float(a.pow(2).mean().sqrt()):.3f})')
    assert dev <= 5e-2


def test_sampler_bf16_deviation():
    """full 100-step DDPM run, same supplied noise: bf16-operand mel vs fp32 mel (normalised units, range [-1, 1]).
    A SELF-COMPARISON: both trajectories come from this library's HIP kernels (the fp32 one is pinned to the oracle elsewhere:
    tests/test_gpu_configs.py); what pins the bf16 arithmetic to an independent restatement is ONE evaluation against the emulating oracle
    (tests/test_gpu_configs.py::test_config2_bf16_full_size_vs_emulating_oracle) and the step tail on small shapes — not this trajectory.
    configs[2] is a throughput configuration."""
    use_config()
    from bisinger_amd.diffnet import DiffNet
    from bisinger_amd.diffusion import GaussianDiffusion
    from bisinger_amd.hparams import hparams

    class Enc:
        def __len__(self):
            return 65

        def pad(self):
            return 0

    m = GaussianDiffusion(Enc(), 80, DiffNet(80), timesteps=100, K_step=100, spec_min=hparams['spec_min'], spec_max=hparams['spec_max'])
    load_formula_weights(m, 0, synth.DIFFNET_GAIN)
    m = m.cuda().eval()
    B, T = 2, 128
    rs = np.random.RandomState(3)
    cond = T_(rs.standard_normal((B, 256, T)).astype(np.float32)).cuda()
    noise = T_(synth.synth_noise(100, B, 80, T, seed=7)).cuda()
    res = {}
    for mode in ('fp32', 'bf16'):
        m.denoise_fn.set_compute(mode)
        x = noise[0][:, None].contiguous().clone()
        res[mode] = m.sample(cond, x, noise=noise[1:], n_steps=100).clone()
    torch.cuda.synchronize()
    m.denoise_fn.set_compute('fp32')
    dev = maxabs(res['fp32'], res['bf16'])
    rms = float((res['fp32'] - res['bf16']).pow(2).mean().sqrt())
    print(f'100-step sampler, bf16 operands vs fp32: max abs {dev:.3e}, rms {rms:.3e} (normalised mel in [-1, 1])')
    assert torch.isfinite(res['bf16']).all()
    assert dev <= 0.1 and rms <= 0.02


def _sampler_model():
    use_config()
    from bisinger_amd.diffnet import DiffNet
    from bisinger_amd.diffusion import GaussianDiffusion
    from bisinger_amd.hparams import hparams

    class Enc:
        def __len__(self):
            return 65

        def pad(self):
            return 0

    m = GaussianDiffusion(Enc(), 80, DiffNet(80), timesteps=100, K_step=100, spec_min=hparams['spec_min'], spec_max=hparams['spec_max'])
    load_formula_weights(m, 0, synth.DIFFNET_GAIN)
    return m.cuda().eval()


@pytest.mark.parametrize('B,T', [(3, 77), (2, 200)])
def test_bf16_step_tail_vs_emulating_oracle(B, T):
    """The bf16 step tail (step_tail_bf16_kernel: skip projection, output projection, p_sample, next input projection on bf16
    MFMAs) inside the fused DDPM loop: 3 steps with supplied noise against the oracle with the same roundings emulated —
    evaluation 1 takes an fp32 input projection (bsg_diffnet's first in-projection), evaluations 2, 3 the tail's bf16 one;
    skip sum rounded once (stack launch); skip / output projection with bf16 operands.  Partial tiles, T < 2 tiles."""
    from oracle import diffusion as odf
    m = _sampler_model()
    sd = cpu_sd(m)
    rs = np.random.RandomState(B * 1000 + T)
    cond = T_(rs.standard_normal((B, 256, T)).astype(np.float32))
    noise = T_(synth.synth_noise(3, B, 80, T, seed=5))
    x0 = noise[0][:, None].contiguous()
    m.denoise_fn.set_compute('bf16')
    try:
        got = m.sample(cond.cuda(), x0.clone().cuda(), noise=noise[1:].cuda(), n_steps=3).cpu()
        path = m.denoise_fn.last_path()
    finally:
        m.denoise_fn.set_compute('fp32')
    ref32 = m.sample(cond.cuda(), x0.clone().cuda(), noise=noise[1:].cuda(), n_steps=3).cpu()
    assert path == 'stack_bf16'
    sch = odf.make_schedule(100, 'linear', 0.06)
    def emulate(tail):
        calls = []

        def denoise(x, t):
            calls.append(1)
            return odn.diffnet_forward(sd, x, t, cond, 'denoise_fn.', operand_bf16=True, skip_rounding='final', tail_bf16=tail,
                                       in_bf16=tail and len(calls) > 1)
        x = x0
        for k, i in enumerate((99, 98, 97)):
            x = odf.p_sample(sch, denoise, x, torch.full((B,), i, dtype=torch.long), noise[1 + k][:, None])
        return x
    x, x_f32tail = emulate(True), emulate(False)
    rms_alt = float((got - x_f32tail).pow(2).mean().sqrt())
    e, q = maxabs(got, x), maxabs(x, ref32)
    rms = float((got - x).pow(2).mean().sqrt())
    rms_q = float((x - ref32).pow(2).mean().sqrt())
    print(f'bf16 tail B={B} T={T}, 3 fused steps: HIP vs bf16-emulating oracle max-abs {e:.2e} rms {rms:.2e} (rms {rms_alt:.2e} against '
          f'the emulation WITHOUT the roundings of the tail); the roundings cost {q:.2e} / {rms_q:.2e} vs the fp32 path')
    # 1-ulp bf16 flips (a 1e-7 summation-order difference ahead of a rounding) are carried forward: statistical agreement, far
    # closer to the emulation than the emulation is to fp32; a wrong weight tile or row order shows as O(1)
    assert rms <= 0.8 * rms_q and e <= q + 1e-4
    assert rms < 0.9 * rms_alt          # the tail's own roundings are what the kernel does


def test_bf16_step_tail_switch_and_plms(monkeypatch):
    """BSG_TAIL_BF16=0 keeps the fp32 tail in the bf16 configuration; both give finite, close results over 12 DDPM steps
    (supplied noise and Philox) and over a PLMS run (the tail's multistep form)."""
    from bisinger_amd.hparams import hparams
    m = _sampler_model()
    B, T = 3, 77
    cond = torch.randn(B, 256, T, generator=torch.Generator().manual_seed(5)).cuda()
    noise = T_(synth.synth_noise(12, B, 80, T, seed=2)).cuda()
    x0 = noise[0][:, None].contiguous()
    m.denoise_fn.set_compute('bf16')
    try:
        res = {}
        for mode in ('1', '0'):
            monkeypatch.setenv('BSG_TAIL_BF16', mode)
            a = m.sample(cond, x0.clone(), noise=noise[1:], n_steps=12).clone()
            b = m.sample(cond, x0.clone(), seed=9, n_steps=12).clone()
            hparams['pndm_speedup'] = 5
            try:
                c = m.sample(cond, x0.clone()).clone()
            finally:
                hparams['pndm_speedup'] = 0
            res[mode] = (a, b, c)
    finally:
        m.denoise_fn.set_compute('fp32')
    for i, name in enumerate(('ddpm, supplied noise', 'ddpm, philox', 'plms/5')):
        a, b = res['1'][i], res['0'][i]
        assert bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all())
        dev, rms = maxabs(a, b), float((a - b).pow(2).mean().sqrt())
        print(f'bf16 tail vs fp32 tail, {name}: max-abs {dev:.2e} rms {rms:.2e}')
        assert not torch.equal(a, b)
        assert dev <= 0.1 and rms <= 0.01


def test_bf16_stack_launch_matches_per_layer_launches(tmp_path):
    """The default bf16 path (BSG_STACK_BF16=0 turns it off): all 20 layers of the bf16 configuration in one launch (x in registers, fp32 skip sum, edges exchanged
    between neighbour tiles) against one launch per layer.  The stack form never rounds the running skip sum to bf16, so
    the two differ by that rounding (not bit-identical); both must sit within bf16 noise of each other, repeat bit for bit,
    and report no hand-off give-ups.  Two launch groups at (40, 640): 10 tiles per row, 25 rows per group."""
    import json
    import os
    import subprocess
    import sys
    code = r'''
import sys, json, hashlib, torch, numpy as np
sys.path.insert(0, %r)
from tests.util import load_formula_weights, use_config
from bisinger_amd import synth
torch.set_grad_enabled(False)
use_config()
from bisinger_amd.diffnet import DiffNet
net = load_formula_weights(DiffNet(80), 0, synth.DIFFNET_GAIN, prefix='denoise_fn.').cuda()
net.set_compute('bf16')
out = {}
for B, T in ((16, 1000), (40, 640), (3, 77)):
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, 1, 80, T, generator=g).cuda(); cond = torch.randn(B, 256, T, generator=g).cuda()
    t = (torch.arange(B) * 7 %% 100).cuda()
    e1 = net(x, t, cond).clone(); e2 = net(x, t, cond).clone()
    out[f'{B}x{T}'] = {'same': bool(torch.equal(e1, e2)), 'path': net.last_path(), 'finite': bool(torch.isfinite(e1).all())}
    np.save(sys.argv[1] + f'.{B}x{T}.npy', e1.cpu().numpy())
out['timeouts'] = net.handoff_timeouts()
print(json.dumps(out))
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    res = {}
    for mode in ('0', '1'):
        env = dict(os.environ, BSG_STACK_BF16=mode)
        o = subprocess.run([sys.executable, '-c', code, str(tmp_path / f'm{mode}')], env=env, capture_output=True, text=True, timeout=600)
        assert o.returncode == 0, o.stderr[-2000:]
        res[mode] = json.loads(o.stdout.strip().splitlines()[-1])
    assert res['1']['timeouts'] == 0
    for k in ('16x1000', '40x640', '3x77'):
        assert res['1'][k]['path'] == 'stack_bf16' and res['0'][k]['path'] == 'bf16'
        assert res['1'][k]['same'] and res['1'][k]['finite'] and res['0'][k]['same']
        a = np.load(str(tmp_path / f'm0.{k}.npy')); b = np.load(str(tmp_path / f'm1.{k}.npy'))
        dev, rms = float(np.abs(a - b).max()), float(np.sqrt(((a - b) ** 2).mean()))
        print(f'bf16 stack vs per-layer {k}: max-abs {dev:.2e}, rms {rms:.2e}')
        assert dev <= 2e-2 and rms <= 2e-3



def test_bf16_trajectory_vs_emulating_oracle():
    """the WHOLE 100-step DDPM trajectory of the bf16-operand configuration (stack launch + bf16 step tail) at B = 2,
    T = 256 against an INDEPENDENT oracle that emulates the same roundings — operands of both GEMMs, the skip sum rounded once per
    evaluation (the stack launch keeps it in fp32 registers), skip / output projection and, from the second evaluation on, the input
    projection with bf16 operands — not against the HIP fp32 path (that comparison is tests/test_gpu_bf16.py::test_bf16_sampler_... and
    test_gpu_fullsize.py, self-comparisons).  Same statistical bounds as test_config2_bf16_full_size_vs_emulating_oracle: bf16 roundings
    sit at decision boundaries, a 1e-7 summation-order difference flips an operand by one bf16 ulp now and then and 100 steps carry every
    flip forward, so the agreement with the emulation is statistical — closer (rms) to the emulation than the emulation is to fp32, a
    max-abs of the order of the roundings' own; a wrong tile, tap, channel order or tail weight shows as O(1)."""
    from oracle import diffusion as odf
    m = _sampler_model()
    sd = cpu_sd(m)
    B, T = 2, 256
    rs = np.random.RandomState(77)
    cond = T_(rs.standard_normal((B, 256, T)).astype(np.float32))
    noise = T_(synth.synth_noise(100, B, 80, T, seed=6))
    x0 = noise[0][:, None].contiguous()
    m.denoise_fn.set_compute('bf16')
    try:
        got = m.sample(cond.cuda(), x0.clone().cuda(), noise=noise[1:].cuda(), n_steps=100).cpu()
        path = m.denoise_fn.last_path()
    finally:
        m.denoise_fn.set_compute('fp32')
    assert path == 'stack_bf16' and bool(torch.isfinite(got).all())
    sch = odf.make_schedule(100, 'linear', 0.06)
    calls = []

    def denoise_emu(x, t):
        calls.append(1)
        return odn.diffnet_forward(sd, x, t, cond, 'denoise_fn.', operand_bf16=True, skip_rounding='final', tail_bf16=True, in_bf16=len(calls) > 1)
    emu = odf.ddpm_sample(sch, denoise_emu, x0, noise[1:][:, :, None], 100)
    f32 = odf.ddpm_sample(sch, lambda x, t: odn.diffnet_forward(sd, x, t, cond, 'denoise_fn.'), x0, noise[1:][:, :, None], 100)
    e, q = maxabs(got, emu), maxabs(emu, f32)
    rms = float((got - emu).pow(2).mean().sqrt())
    rms_q = float((emu - f32).pow(2).mean().sqrt())
    rms_x = float(f32.pow(2).mean().sqrt())
    print(f'bf16 trajectory B={B} T={T}, 100 steps (x rms {rms_x:.3f}): HIP-bf16 vs bf16-emulating oracle max-abs {e:.3e}, rms {rms:.2e}; the roundings '
          f'themselves cost max-abs {q:.3e}, rms {rms_q:.2e} vs the fp32 oracle')
    assert rms <= rms_q and e <= 2.0 * q and e <= 0.1 * max(rms_x, 0.1)


def test_set_compute_between_prepare_and_forward_is_refused_both_ways():
    """a BF16 prepare writes the bf16 quads of the conditioner term only — no fp32 copy since round 5 — so a direct ABI
    user who calls bsg_diffnet_set_compute(F32) behind it and then evaluates would read a stale / uninitialised fp32 term.  check_bound now
    refuses a compute mode other than the prepared one in BOTH directions (BSG_ESTATE + a message); a new prepare makes the handle usable again
    (the Python wrapper does that by itself: DiffNet.set_compute drops the binding)."""
    from ctypes import c_void_p
    from bisinger_amd import _lib
    m = _sampler_model()
    net = m.denoise_fn
    lib = _lib.load()
    B, T = 2, 96
    rs = np.random.RandomState(8)
    cond = T_(rs.standard_normal((B, 256, T)).astype(np.float32)).cuda()
    x = T_(rs.standard_normal((B, 80, T)).astype(np.float32)).cuda()
    t = torch.tensor([3, 70], device='cuda')
    eps = torch.empty_like(x)
    h = net.handle()
    fwd = lambda: lib.bsg_diffnet_forward(h, _lib.ptr(x), _lib.ptr(t), _lib.ptr(eps), B, T, _lib.stream_ptr())
    prep = lambda: lib.bsg_diffnet_prepare(h, _lib.ptr(cond), B, T, _lib.stream_ptr())
    try:
        for first, second in ((1, 0), (0, 1)):          # BSG_COMPUTE_BF16 = 1, BSG_COMPUTE_F32 = 0
            assert lib.bsg_diffnet_set_compute(h, first) == 0 and prep() == 0 and fwd() == 0
            assert lib.bsg_diffnet_set_compute(h, second) == 0
            rc = fwd()
            assert rc != 0 and 'bsg_diffnet_prepare again' in lib.bsg_last_error().decode(), lib.bsg_last_error()
            assert prep() == 0 and fwd() == 0                  # bound again under the new mode: valid
            torch.cuda.synchronize()
            assert bool(torch.isfinite(eps).all())
    finally:
        lib.bsg_diffnet_set_compute(h, 0)
        net._bound = None
        net.compute_dtype = 'fp32'
